"""Parity of the fused HIP flow stack (through the C ABI) against the golden vectors captured
from the reference's PyTorch modules and against the CPU oracle.

Tolerance (north star: <= 1e-4 rel fp32).  Two measures, both asserted for every split precision:
    norm-wise    rel(got, ref) = max|got - ref| / max|ref|            <= REL[prec]
    elementwise  |got - ref| <= 1e-4 * |ref| + ATOL_SCALE[prec] * max|ref|   (np.testing.assert_allclose with
                 the absolute term tied to the tensor's scale: entries near zero cannot be held to a relative bound)
f16x3 (default: fp16 hi/lo operands, 22 significant bits) and bf16x6 are fp32-class; bf16x3 (16 bits) meets 1e-4
norm-wise; elementwise it is held only to atol = 1e-4 * scale (its 16-bit parts leave ~3e-5 of the tensor's scale on
the small early-layer logvar tensors) -- which is why it is no longer the default.  Plain bf16 (one MFMA product) does NOT meet 1e-4 and
is checked only against its own, looser, measured bound."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle.gen_golden import layer_inputs

pytestmark = pytest.mark.gpu

REL = {"bf16x6": 2e-6, "f16x3": 4e-6, "bf16x3": 1e-4, "bf16": 5e-3}
ATOL_SCALE = {"bf16x6": 1e-6, "f16x3": 2e-6, "bf16x3": 1e-4}


def assert_elementwise(got, ref, prec, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=ATOL_SCALE[prec] * float(np.abs(ref).max()), err_msg=str(what))


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


def rel(got, ref):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got.astype(np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.fixture(params=["tile32", "tile16"])
def tiling(request):
    """Run a test once per tiling of the fused eval kernel: 32-point tiles (csrc/flow.hip) and, forced, 16-point tiles
    (csrc/flow16.hip: the small-batch variant; by default it serves B * ceil(N / 16) <= 1024 at f16x3)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd._lib import lib
    old = lib().dpf_flow_set_tile16(1 if request.param == "tile16" else 0)
    yield request.param
    lib().dpf_flow_set_tile16(old)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz")), json.load(open(os.path.join(golden_dir, name + ".json")))


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3", "bf16x6", "bf16"])
def test_single_layer_vs_reference_golden(golden_dir, prec, tiling):
    if tiling == "tile16" and prec != "f16x3":
        pytest.skip("16-point tiles exist for f16x3 only")
    nets = _gpu()
    gold, meta = _load(golden_dir, "flow_layer")
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    for case in meta["cases"]:
        if case["bn"] != "eval":
            continue
        mod = nets.CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
        mod.load_state_dict(FO.to_torch(FO.make_layer_state(case["seed"], F, G, case["warp"])), strict=True)
        mod = mod.cuda().eval()
        mod.precision = prec
        p, g, _, _, _ = layer_inputs(case["seed"], B, N, G)
        with torch.no_grad():
            po, mu, lv = mod(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda(), mode=case["mode"])
        t = case["tag"]
        for name, got in (("p_out", po), ("mu", mu), ("logvar", lv)):
            r = rel(got, gold[t + "/" + name])
            assert r <= REL[prec], (t, name, prec, r)
            if prec in ATOL_SCALE:
                assert_elementwise(got, gold[t + "/" + name], prec, (t, name))
        # channels that are not warped carry exactly mu = 0, logvar = 0 (flows.py:96-97)
        keep = [c for c in range(3) if c not in case["warp"]]
        assert (mu[:, keep] == 0).all() and (lv[:, keep] == 0).all()
        if prec == "bf16x6":
            np.testing.assert_allclose(po.cpu().numpy(), gold[t + "/p_out"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3", "bf16x6"])
def test_decoder_vs_reference_golden(golden_dir, prec, tiling):
    if tiling == "tile16" and prec != "f16x3":
        pytest.skip("16-point tiles exist for f16x3 only")
    nets = _gpu()
    gold, meta = _load(golden_dir, "flow_decoder")
    for case in meta["cases"]:
        if case.get("bn") == "train":
            continue
        c, nf, B, N, G, seed, mode = (case[k] for k in ("tag", "n_flows", "B", "N", "G", "seed", "mode"))
        dec = nets.LocalCondRNVPDecoder(nf, 64, G)
        dec.load_state_dict(FO.to_torch(FO.make_decoder_state(seed, nf, 64, G)), strict=True)
        dec = dec.cuda().eval()
        dec.precision = prec
        tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
        src = torch.from_numpy(z if mode == "direct" else tgt).cuda()
        tg = torch.from_numpy(g).cuda()
        with torch.no_grad():
            ps, mus, lvs = dec(src, tg, mode=mode)
        assert len(ps) == len(mus) == len(lvs) == 3 * nf                       # decoders.py:54-72
        for k in case["picks"]:
            for name, lst in (("ps", ps), ("mus", mus), ("logvars", lvs)):
                r = rel(lst[k], gold["%s/%s%d" % (c, name, k)])
                assert r <= REL[prec], (c, name, k, prec, r)
                assert_elementwise(lst[k], gold["%s/%s%d" % (c, name, k)], prec, (c, name, k))
        assert rel(lvs.total(), gold[c + "/sum_logvars"]) <= REL[prec]
        assert rel(sum(lvs), gold[c + "/sum_logvars"]) <= REL[prec]            # python sum over the list views
        # the loss exactly as models.py:152-171 + losses.py:11-15 assemble it
        prior_mu = torch.zeros(B, 3, N, device="cuda")
        prior_lv = torch.full((B, 3, N), -3.6, device="cuda")
        smp = ps + [src] if mode == "inverse" else [src] + ps
        nll = nets.PointFlowNLL()(smp, [prior_mu] + mus, [prior_lv] + lvs)
        np.testing.assert_allclose(float(nll), float(gold[c + "/nll"]), rtol=1e-4 if prec == "bf16x3" else 2e-5)
        assert_elementwise(lvs.total(), gold[c + "/sum_logvars"], prec, (c, "sum_logvars"))
        # `+=` of models.py:119-122 extends a python list with the FlowList
        acc = [prior_lv]
        acc += lvs
        assert len(acc) == 3 * nf + 1 and acc[1] is lvs[0]


def test_l14_truncated_stack_vs_reference_golden(golden_dir, tiling):
    """The BASELINE metric's L=14 = first 14 direct-order layers of n_flows=5."""
    nets = _gpu()
    gold, _ = _load(golden_dir, "flow_decoder")
    dec = nets.LocalCondRNVPDecoder(5, 64, 128)
    dec.load_state_dict(FO.to_torch(FO.make_decoder_state(7, 5, 64, 128)), strict=True)
    dec = dec.cuda().eval()
    tgt, z, g = FO.synthetic_inputs(7, 2, 128, 128)
    for lists in (True, False):
        dec.materialize_lists = lists
        with torch.no_grad():
            ps, mus, lvs = dec(torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda(), mode="direct", n_layers=14)
        assert len(ps) == (14 if lists else 1)
        assert rel(ps[-1], gold["nf5_L14_direct/final"]) <= REL["f16x3"]           # default precision
        assert rel(lvs.total(), gold["nf5_L14_direct/sum_logvars"]) <= REL["f16x3"]
        assert_elementwise(ps[-1], gold["nf5_L14_direct/final"], "f16x3", "L14 final")


@pytest.mark.parametrize("shape", [(1, 1, 128), (3, 31, 128), (2, 33, 128), (2, 129, 512), (5, 257, 128), (2, 2500, 512)])
def test_ragged_sizes_vs_oracle(shape, tiling):
    """N not a multiple of the 32- / 16-point tile or of the workgroup's points, B = 1, G = 512."""
    nets = _gpu()
    B, N, G = shape
    nf = 2
    state = FO.make_decoder_state(40 + N, nf, 64, G)
    dec = nets.LocalCondRNVPDecoder(nf, 64, G)
    dec.load_state_dict(FO.to_torch(state), strict=True)
    dec = dec.cuda().eval()
    tgt, z, g = FO.synthetic_inputs(40 + N, B, N, G)
    for mode, src in (("direct", z), ("inverse", tgt)):
        with torch.no_grad():
            ps, mus, lvs = dec(torch.from_numpy(src).cuda(), torch.from_numpy(g).cuda(), mode=mode)
            rps, rmus, rlvs = FO.decoder(FO.to_torch(state), nf, torch.from_numpy(src), torch.from_numpy(g), mode)
        for k in range(3 * nf):
            assert rel(ps[k], rps[k]) <= REL["f16x3"] and rel(mus[k], rmus[k]) <= REL["f16x3"]
            assert rel(lvs[k], rlvs[k]) <= REL["f16x3"], (shape, mode, k)
        assert torch.isfinite(ps.stacked).all()


@pytest.mark.parametrize("shrink", [2.0 ** -6, 2.0 ** -10], ids=["W1x2^-6", "W1x2^-10"])
def test_f16x3_small_weights_vs_float64(shrink, tiling):
    """VERDICT r04 #5 / ADVICE r03 #3: f16x3 packs the 64 x 64 SharedDot's weights as fp16 hi + fp16 lo.  Unscaled, the lo parts of
    init-scale weights (|W1| ~ 0.04) are fp16 subnormals, and a branch whose W1 is small altogether loses bits of the hi part
    too (at 2^-10 of the init scale: ~9 significant bits).  r05 packs W1 times a power of two per (layer, branch) -- largest
    entry in [2^13, 2^14) -- and the FiLM block carries the inverse (csrc/flow_common.h: w1_pow2_scale).  Here every W1 of a
    decoder is shrunk by 2^-6 / 2^-10 (the BatchNorm behind it follows, so the layer stays a live function of W1) and the
    stack must still sit within the f16x3 bar of float64, in both tilings."""
    nets = _gpu()
    B, N, G, nf = 3, 500, 128, 2
    sd = FO.to_torch(FO.make_decoder_state(77, nf, 64, G))
    for k in list(sd):
        if k.endswith("sd1.weight"):
            sd[k] = sd[k] * shrink
        if k.endswith("sd1_bn.running_mean"):
            sd[k] = sd[k] * shrink
        if k.endswith("sd1_bn.running_var"):
            sd[k] = sd[k] * (shrink * shrink)
    dec = nets.LocalCondRNVPDecoder(nf, 64, G)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().eval()
    ref = nets.LocalCondRNVPDecoder(nf, 64, G)
    ref.load_state_dict(sd, strict=True)
    ref = ref.cuda().double().eval()
    tgt, z, g = FO.synthetic_inputs(77, B, N, G)
    for mode, src in (("direct", z), ("inverse", tgt)):
        with torch.no_grad():
            ps, mus, lvs = dec(torch.from_numpy(src).cuda(), torch.from_numpy(g).cuda(), mode=mode)
            rps, rmus, rlvs = ref.forward_torch(torch.from_numpy(src).cuda().double(), torch.from_numpy(g).cuda().double(), mode=mode)
        worst = max(max(rel(ps[k], rps[k]), rel(mus[k], rmus[k]), rel(lvs[k], rlvs[k])) for k in range(3 * nf))
        assert worst <= REL["f16x3"], (shrink, mode, worst)


@pytest.mark.parametrize("B,G", [(32, 128), (64, 512), (32, 512)], ids=["configs1_B32_G128", "configs2_B64_G512", "configs3_B32_G512"])
def test_full_size_properties(B, G, tiling):
    """(once per tiling of the eval kernel: the per-point independence below is bit-exact INSIDE a tiling; across tilings
    the matrix cores add their 16 / 32 K-slots in a different order -- test_tile16_* holds the two to the oracle.)
    BASELINE.json configs[1] (B=32, G=128), configs[2] (all-classes model: B=64, G=512) and configs[3] (SVR decoder
    shapes: B=32, G=512), all N=2048 and 63 layers, at FULL size: direct then inverse returns the input up to the
    reference's own sqrt(1+eps) keep-channel drift; outputs independent of batch composition; two clouds against the oracle."""
    nets = _gpu()
    N, nf = 2048, 21
    state = FO.make_decoder_state(5, nf, 64, G)
    dec = nets.LocalCondRNVPDecoder(nf, 64, G)
    dec.load_state_dict(FO.to_torch(state), strict=True)
    dec = dec.cuda().eval()
    dec.precision = "bf16x6"
    tgt, z, g = FO.synthetic_inputs(5, B, N, G)
    tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
    with torch.no_grad():
        ps, mus, lvs = dec(tz, tg, mode="direct")
        back, _, lvs_b = dec(ps[-1], tg, mode="inverse")
        # eval-mode layers are a pure per-point map: a sub-batch gives bit-identical results
        ps_sub, _, _ = dec(tz[5:9].contiguous(), tg[5:9].contiguous(), mode="direct")
        # cross-check a few clouds against the CPU oracle at full depth
        rps, _, rlvs = FO.decoder(FO.to_torch(state), nf, torch.from_numpy(z[:2]), torch.from_numpy(g[:2]), "direct")
    assert torch.equal(ps_sub[-1], ps[-1][5:9])
    assert rel(ps[-1][:2], rps[-1]) <= 1e-5 and rel(lvs.total()[:2], sum(rlvs)) <= 1e-5
    # round trip: the inverse conditions on the OUTPUT-side keep channels (flows.py:106), which differ from
    # the direct pass's inputs by sqrt(1+1e-6) per layer, so the round trip is exact only to ~63 * 1e-6
    assert rel(back[0], z) < 5e-4
    assert len(ps) == 63 and ps.stacked.shape == (63, B, 3, N)
    # bf16x3 at this size takes the variant with TWO layers per LDS buffer (a barrier every other layer; 63 layers: the
    # last buffer holds one), a 4-cloud sub-batch the one-layer-per-buffer 4-wave variant: same bits, and the oracle
    for prec in ("bf16x3", "f16x3"):
        dec.precision = prec
        with torch.no_grad():
            ps3, _, lvs3 = dec(tz, tg, mode="direct")
            ps3_sub, _, _ = dec(tz[5:9].contiguous(), tg[5:9].contiguous(), mode="direct")
            inv3, _, _ = dec(ps3[-1], tg, mode="inverse")
            inv3_sub, _, _ = dec(ps3[-1][5:9].contiguous(), tg[5:9].contiguous(), mode="inverse")
        assert torch.equal(ps3_sub[-1], ps3[-1][5:9]) and torch.equal(ps3_sub[31], ps3[31][5:9])
        assert torch.equal(inv3_sub[0], inv3[0][5:9])
        assert rel(ps3[-1][:2], rps[-1]) <= REL[prec] and rel(lvs3.total()[:2], sum(rlvs)) <= REL[prec]
        assert_elementwise(ps3[-1][:2], rps[-1], prec, "full size, 63 layers: points")
        assert_elementwise(lvs3.total()[:2], sum(rlvs), prec, "full size, 63 layers: sum of logvars")


@pytest.mark.parametrize("B", [4, 8])
def test_tile16_serves_a_rank_sized_batch_by_default_and_matches_oracle_and_tile32(B):
    """a rank of an 8-GPU job holds 4 (BASELINE's B = 32) or 8 (configs[2]: B = 64) clouds of 2048 points.
    Such a batch is served by the 16-point-tile kernel WITHOUT being asked (dpf_flow_tile16_launches moves); its results
    meet the oracle at the 32-point kernel's bars -- norm-wise and elementwise, direct and inverse, L = 14 and the real depth
    L = 63, with the per-layer lists -- and agree with the 32-point kernel's to rounding."""
    nets = _gpu()
    from dpf_nets_amd._lib import lib
    N, G, nf = 2048, 128, 21
    state = FO.make_decoder_state(9, nf, 64, G)
    dec = nets.LocalCondRNVPDecoder(nf, 64, G)
    dec.load_state_dict(FO.to_torch(state), strict=True)
    dec = dec.cuda().eval()
    tgt, z, g = FO.synthetic_inputs(9, B, N, G)
    tz, tt, tg = torch.from_numpy(z).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    before = lib().dpf_flow_tile16_launches()
    with torch.no_grad():
        ps, mus, lvs = dec(tz, tg, mode="direct")
        p14, _, lv14 = dec(tz, tg, mode="direct", n_layers=14)
        inv, imus, ilvs = dec(tt, tg, mode="inverse")
    assert lib().dpf_flow_tile16_launches() == before + 3                     # all three calls took the 16-point kernel
    old = lib().dpf_flow_set_tile16(0)
    try:
        with torch.no_grad():
            ps32, mus32, lvs32 = dec(tz, tg, mode="direct")
            inv32, _, ilvs32 = dec(tt, tg, mode="inverse")
    finally:
        lib().dpf_flow_set_tile16(old)
    assert lib().dpf_flow_tile16_launches() == before + 3
    with torch.no_grad():
        rps, rmus, rlvs = FO.decoder(FO.to_torch(state), nf, torch.from_numpy(z[:2]), torch.from_numpy(g[:2]), "direct")
        rinv, _, rilvs = FO.decoder(FO.to_torch(state), nf, torch.from_numpy(tgt[:2]), torch.from_numpy(g[:2]), "inverse")
    for k in (0, 13, 31, 62):
        for name, got, ref in (("ps", ps, rps), ("mus", mus, rmus), ("logvars", lvs, rlvs)):
            assert rel(got[k][:2], ref[k]) <= REL["f16x3"], (name, k)
            assert_elementwise(got[k][:2], ref[k], "f16x3", (name, k))
    assert rel(p14[-1][:2], rps[13]) <= REL["f16x3"] and rel(lv14.total()[:2], sum(rlvs[:14])) <= REL["f16x3"]
    assert rel(lvs.total()[:2], sum(rlvs)) <= REL["f16x3"]
    assert rel(inv[0][:2], rinv[0]) <= REL["f16x3"] and rel(ilvs.total()[:2], sum(rilvs)) <= REL["f16x3"]
    assert_elementwise(inv[0][:2], rinv[0], "f16x3", "inverse, 63 layers")
    # the two tilings against each other, every cloud
    assert rel(ps[-1], ps32[-1].cpu().numpy()) <= 2 * REL["f16x3"] and rel(inv[0], inv32[0].cpu().numpy()) <= 2 * REL["f16x3"]
    assert rel(lvs.total(), lvs32.total().cpu().numpy()) <= 2 * REL["f16x3"]
    # a per-point map: a sub-batch of two clouds (served by the same kernel) gives the same bits
    with torch.no_grad():
        sub, _, _ = dec(tz[1:3].contiguous(), tg[1:3].contiguous(), mode="direct")
    assert torch.equal(sub[-1], ps[-1][1:3]) and torch.equal(sub[20], ps[20][1:3])


def test_f16x3_range_guard_falls_back_on_a_collapsed_batchnorm_variance():
    """VERDICT r02 missing #6 / ADVICE: f16x3's hi/lo split is exact below 2048 and saturates at 65504.  A checkpoint whose
    BN0 running variance collapsed (1e-12: the folded first layer gains a factor 316) must not be evaluated at f16x3
    silently: the pack-time bound trips, a warning is issued and the stack runs at bf16x6 (fp32-class, no range limit) --
    finite, and equal to the tensor-op path; a healthy checkpoint keeps f16x3."""
    nets = _gpu()
    import warnings
    state = FO.to_torch(FO.make_decoder_state(61, 2, 64, 128))
    tgt, z, g = FO.synthetic_inputs(61, 3, 200, 128)
    tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
    dec = nets.LocalCondRNVPDecoder(2, 64, 128)
    dec.load_state_dict(state, strict=True)
    dec = dec.cuda().eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # healthy weights: no warning, f16x3
        with torch.no_grad():
            dec(tz, tg, mode="direct")
    assert dec.stack().last_precision == "f16x3"
    bad = {k: v.clone() for k, v in state.items()}
    for k in bad:
        if k.endswith("sd0_bn.running_var"):
            bad[k].fill_(1e-12)
    dec2 = nets.LocalCondRNVPDecoder(2, 64, 128)
    dec2.load_state_dict(bad, strict=True)
    dec2 = dec2.cuda().eval()
    with pytest.warns(UserWarning, match="f16x3 is outside its exact range"):
        with torch.no_grad():
            ps, mus, lvs = dec2(tz, tg, mode="direct")
    assert dec2.stack().last_precision == "bf16x6"
    with torch.no_grad():
        rps, _, rlvs = dec2.forward_torch(tz, tg, mode="direct")
    assert torch.isfinite(ps[-1]).all()
    assert rel(ps[-1], rps[-1]) <= 1e-4 and rel(lvs.total(), sum(rlvs)) <= 1e-4     # under a x316 first layer


def test_f16x3_large_activations_stay_fp32_class_below_the_limit():
    """a case with LARGE hidden activations.  gamma0 scaled so that |h0| reaches the hundreds (bound < 2048: the
    guard keeps f16x3): the fp16 hi + lo split must still be fp32-class against the tensor-op path."""
    nets = _gpu()
    state = FO.to_torch(FO.make_decoder_state(62, 2, 64, 128))
    for k in state:
        if k.endswith("sd0_bn.weight") or k.endswith("sd0_bn.bias"):
            state[k] = state[k] * 40.0                                       # gamma0, beta0
        if k.endswith("_sd1.weight"):
            state[k] = state[k] / 40.0                                       # keep the layer's output scale
    dec = nets.LocalCondRNVPDecoder(2, 64, 128)
    dec.load_state_dict(state, strict=True)
    dec = dec.cuda().eval()
    tgt, z, g = FO.synthetic_inputs(62, 3, 300, 128)
    tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
    with torch.no_grad():
        ps, mus, lvs = dec(tz, tg, mode="direct")
        rps, _, rlvs = dec.forward_torch(tz.double().float(), tg, mode="direct")
    st = dec.stack()
    assert st.last_precision == "f16x3" and 100.0 < st._f16_bound[0] < 2048.0, st._f16_bound
    assert rel(ps[-1], rps[-1]) <= 1e-5 and rel(lvs.total(), sum(rlvs)) <= 1e-5


def test_module_semantics():
    nets = _gpu()
    dec = nets.LocalCondRNVPDecoder(1, 64, 128).cuda().eval()
    p = torch.randn(2, 3, 64, device="cuda") * 0.3
    g = torch.randn(2, 128, device="cuda")
    with torch.no_grad():
        a, _, _ = dec(p, g, mode="direct")
        # in-place weight update (what an optimizer step does) must invalidate the packed weights
        dec.flows[0].nvp1.T_mu_1[1].weight.add_(0.05)
        dec.flows[0].nvp1.T_mu_0[3].weight.mul_(1.0)
        b, _, _ = dec(p, g, mode="direct")
        assert not torch.equal(a[-1], b[-1])
        # load_state_dict also invalidates
        sd = {k: v.clone() for k, v in dec.state_dict().items()}
        sd["flows.0.nvp2.T_mu_1.mu_sd2.bias"] += 0.25
        dec.load_state_dict(sd)
        c, mus, _ = dec(p, g, mode="direct")
        assert not torch.equal(b[-1], c[-1])
        with pytest.raises(ValueError):
            dec(p, g, mode="sideways")
        with pytest.raises(RuntimeError):
            dec(p.cpu(), g.cpu())                  # no CPU fallback for the fused path
        with pytest.raises(RuntimeError):
            dec(p, g[:, :64].contiguous())
    # training mode: tensor-op path with batch statistics, differentiable, same interface
    dec.train()
    pr = p.clone().requires_grad_(True)
    ps, mus, lvs = dec(pr, g, mode="inverse")
    assert isinstance(ps, list) and len(ps) == 3
    nets.PointFlowNLL()(ps + [pr], [torch.zeros_like(p)] + mus, [torch.zeros_like(p)] + lvs).backward()
    assert pr.grad is not None and torch.isfinite(pr.grad).all()


def test_eval_mode_gradients_vs_reference_golden(golden_dir):
    """CondRealNVPFlow3D.forward is differentiable in eval() mode in the reference (flows.py:95-117).  The fused
    HIP stacks have a backward pass for training-mode BatchNorm only, so an eval-mode call whose inputs require grad is served by
    the reference's op sequence on ATen -- LOUDLY (networks.flows.EvalModeAutogradWarning, once per call site) -- and its outputs
    and input gradients are the reference's (golden cases bn == "eval": grad_p, grad_g of a fixed linear functional)."""
    import warnings
    nets = _gpu()
    from dpf_nets_amd.networks.flows import EvalModeAutogradWarning
    from oracle.gen_golden import layer_inputs as li
    gold, meta = _load(golden_dir, "flow_layer")
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    seen = 0
    for case in meta["cases"]:
        if case["bn"] != "eval":
            continue
        mod = nets.CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
        mod.load_state_dict(FO.to_torch(FO.make_layer_state(case["seed"], F, G, case["warp"])), strict=True)
        mod = mod.cuda().eval()
        p, g, r1, r2, r3 = li(case["seed"], B, N, G)
        tp = torch.from_numpy(p).cuda().requires_grad_(True)
        tg = torch.from_numpy(g).cuda().requires_grad_(True)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            po, mu, lv = mod(tp, tg, mode=case["mode"])
        assert any(issubclass(w.category, EvalModeAutogradWarning) for w in caught), case["tag"]
        ((po * torch.from_numpy(r1).cuda()).sum() + (lv * torch.from_numpy(r2).cuda()).sum() + (mu * torch.from_numpy(r3).cuda()).sum()).backward()
        t = case["tag"]
        for name, got in (("p_out", po), ("mu", mu), ("logvar", lv), ("grad_p", tp.grad), ("grad_g", tg.grad)):
            assert rel(got, gold[t + "/" + name]) <= 2e-5, (t, name, rel(got, gold[t + "/" + name]))
        # ... and the same call under no_grad takes the fused stack: no warning, the same outputs at the fused path's precision
        with warnings.catch_warnings(record=True) as caught, torch.no_grad():
            warnings.simplefilter("always")
            po2, _, _ = mod(tp, tg, mode=case["mode"])
        assert not any(issubclass(w.category, EvalModeAutogradWarning) for w in caught)
        assert rel(po2, gold[t + "/p_out"]) <= REL["f16x3"]
        seen += 1
    assert seen == 12
    # the decoder (the stack of layers) and the fused sampling entry point follow the same rule
    dec = nets.LocalCondRNVPDecoder(1, 64, 128).cuda().eval()
    pr = (torch.randn(2, 3, 64, device="cuda") * 0.3).requires_grad_(True)
    gg = torch.randn(2, 128, device="cuda")
    with pytest.warns(EvalModeAutogradWarning):
        ps, mus, lvs = dec(pr, gg, mode="inverse")
    ps[0].sum().backward()
    assert pr.grad is not None and torch.isfinite(pr.grad).all()


def test_pointflow_nll_fused_reduction():
    """PointFlowNLL (losses.py:11-15) on the fused stack's lists in evaluation: one pass over the cloud and the kernel's
    sum of log-variances with the base distribution's stride-0 expansions read in place (csrc/nll.hip), against the
    reference's arithmetic on the same tensors in float64 and the tensor-op formula in fp32."""
    nets = _gpu()
    import math
    torch.manual_seed(4)
    B, N, G = 5, 777, 128
    dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().eval()
    p = torch.randn(B, 3, N, device="cuda") * 0.3
    g = torch.randn(B, G, device="cuda")
    pm = torch.zeros(1, 3, 1, device="cuda").expand(B, 3, N)                    # models.py:112-117: stride-0 expansions
    pl = (torch.randn(B, 3, 1, device="cuda") * 0.1 - 3.0).expand(B, 3, N)
    nll = nets.PointFlowNLL()
    with torch.no_grad():
        ps, mus, lvs = dec(p, g, mode="inverse")
        calls = []
        from dpf_nets_amd._lib import lib
        orig = lib().dpf_pointflow_nll
        lib().dpf_pointflow_nll = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            got = nll(ps + [p], [pm] + mus, [pl] + lvs)
        finally:
            lib().dpf_pointflow_nll = orig
        assert calls, "the fused reduction was not taken"
        s0 = ps[0].double()
        tot = sum(v.double() for v in lvs) + pl.double() + (s0 - pm.double()) ** 2 / torch.exp(pl.double())
        want = 0.5 * (tot.sum() / B + math.log(2.0 * math.pi) * 3 * N)
        plain = 0.5 * ((sum([pl] + [v.clone() for v in lvs]) + (ps[0] - pm) ** 2 / torch.exp(pl)).sum() / B + math.log(2.0 * math.pi) * 3 * N)
    assert got.shape == () and got.dtype == torch.float32
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (float(got), float(want))
    assert abs(float(got) - float(plain)) <= 1e-5 * abs(float(want))
    assert torch.equal(got, nll(ps + [p], [pm] + mus, [pl] + lvs))                # deterministic
    # a differentiable base distribution goes through the tensor ops and autograd
    plg = pl.clone().requires_grad_(True)
    out = nll(ps + [p], [pm] + mus, [plg] + lvs)
    out.backward()
    assert plg.grad is not None and abs(float(out.detach()) - float(want)) <= 1e-5 * abs(float(want))


def test_steady_state_evaluation_step_copies_nothing_between_host_and_device():
    """The evaluation hot path as a caller of the mirror uses it -- decoder forward (direct, eval-BN) + nn_distance + the CD
    reduction, eager launches -- must not copy between host and device or read a scalar back once the weights are packed
    (the range guard of the fp16 operands reads one scalar per WEIGHT VERSION, not per call)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from torch.profiler import profile, ProfilerActivity
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.metrics.StructuralLosses.nn_distance import nn_distance
    B, N, G = 8, 1024, 128
    dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().eval()
    tgt, z, g = FO.synthetic_inputs(5, B, N, G)
    tz, tg, tt = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda(), torch.from_numpy(np.ascontiguousarray(tgt.transpose(0, 2, 1))).cuda()

    def step():
        with torch.no_grad():
            ps, mus, lvs = dec(tz, tg, mode="direct")
            d1, d2 = nn_distance(ps[-1].transpose(1, 2).contiguous(), tt)
            return (d1.mean(1) + d2.mean(1)).mean()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    keys = {e.key: e.count for e in prof.key_averages()}
    bad = {k: c for k, c in keys.items() if "HtoD" in k or "Host -> Device" in k or "DtoH" in k or
           k in ("aten::index", "aten::item", "aten::_local_scalar_dense", "aten::nonzero")}
    assert not bad, bad
