"""Parity of the HIP TRAINING path (csrc/flow_train.hip through the C ABI: batch-statistics
BatchNorm forward + the whole backward pass) against

  * the golden vectors captured from the reference's PyTorch modules in train() mode
    (outputs, d/dp, d/dg, projections of every parameter gradient, BatchNorm running statistics),
  * the tensor-op restatement of flows.py:95-117 (`forward_torch`) run on the same GPU with
    autograd, at sizes with ragged tiles and many workgroups.

Tolerance: ONE layer: outputs rel <= 1e-4 of the tensor's scale (north star; measured 1e-5..5e-5),
gradients rel <= 1e-3 of the gradient tensor's scale (measured ~2e-5; they pass through two BatchNorm
backward reductions over all B*N points).  A STACK of training-mode layers amplifies any perturbation
of a layer's output by ~1.3x per layer -- fp32 itself goes from 5e-7 after one layer to 2.3e-6 after
six against float64 (tests/diag/train_dbg2.py).  At the default training precision (bf16x6) the 6-layer
stacks are nevertheless held to the one-layer bars, 1e-4 on outputs and 1e-3 on gradients (measured against
float64: outputs 2e-6..8e-6, gradients 2e-5..4e-4, profiles/r02_train_error_budget.json); the opt-in bf16x3 gets
5e-4 / 2e-3."""
import json
import os

import math

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle.gen_golden import layer_inputs, _grad_projection

pytestmark = pytest.mark.gpu

OUT_REL = 1e-4
# r04: the DEFAULT precision, f16x3, runs its gradient contractions (dh0 = W1^T dh1, dW1 = dh1 h0^T) on fp16 hi/lo
# operands with power-of-two scaling -- 22 significant bits -- and is held to 2e-4 on every gradient (the fp32 tensor-op path
# itself sits at 2e-5 .. 2e-4 of float64 on these cases, profiles/r04_train_error_budget.json).  The opt-in precisions keep
# bf16 hi/lo gradient operands (16 bits: bf16x6's third forward part leaves pass 2 no LDS for a third W1^T part) and r03's bars.
GRAD_REL = {"f16x3": 2e-4, "bf16x6": 1e-3, "bf16x3": 3e-3}
STACK_TOL = {"bf16x6": (1e-4, 1e-3), "f16x3": (1e-4, 2e-4), "bf16x3": (5e-4, 2e-3)}     # six layers: (outputs, gradients)
# parameter gradients that are heavily cancelling sums over all points are held to a multiple of what the fp32 tensor-op path
# itself achieves against float64 on the same inputs
R32_FACTOR = {"f16x3": 5.0, "bf16x6": 15.0, "bf16x3": 15.0}     # (measured worst, the nvp2 mu bias at (8, 2048, direct): 4.1 x; r03: 10.3 x)
# Training-mode BatchNorm couples all P = B * N points: ONE ReLU whose pre-activation is within the forward error of zero and
# switches the other way moves every batch statistic, hence every gradient of the stack, by O(1 / P) of its scale (the fp32
# tensor ops flip too, only other ReLUs).  Elementwise bars therefore cannot be held below ~KINK / P on small batches:
# tolerance = max(stated tolerance, KINK / P) -- 2e-4 from P = 20 000 points on; the goldens' own cases have no such ReLU.
KINK = 4.0
# ... and a PARAMETER gradient is a sum over all points whose terms cancel (to 1e-2 .. 1e-3 of their magnitudes for the biases
# of the later layers), which amplifies the O(1 / P) shift of its terms: measured worst (8 x 2048 points, nvp3's BN0 bias)
# 1.1e-3 = 18 / P with the fp32 tensor ops at 1e-6 -- they flip other ReLUs, or none.  The accuracy of the gradient
# CONTRACTIONS is what test_training_stack_error_budget_vs_float64 holds to 3 x the fp32 path's own error; this test holds
# every shape (ragged tiles, > 32 clouds, tiny batches) to what a flipped ReLU allows.  (r05: 25 M pre-activations at (8, 2048) with a
# forward error of ~5e-7 of their scale make ~10 such ReLUs per pass at ANY seed; which ones, and how much gradient hangs on them,
# changes with the forward's rounding: the W1 power-of-two scaling of the training forward drew 1.954e-3 = 32.02 / P on a FiLM
# net's BatchNorm weight -- a sum over 8 clouds normalised over those 8 -- where the unscaled forward had drawn 18 / P: hence 48.)
KINK_SUM = 48.0
# ... and a row of the gradient w.r.t. the per-cloud condition g sums over the N points of its own cloud only
KINK_CLOUD = 2.0
KINK_CAP, KINK_SUM_CAP = 2e-3, 5e-3


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


def rel(got, ref, floor=0.0):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got.astype(np.float64) - ref).max() / (max(float(np.abs(ref).max()), floor) + 1e-30))


def bias_floor(grads, k):
    """A Linear's bias gradient sums the same per-point terms as its weight gradient, without the O(1) activation factor.  Where
    that sum cancels (nvp3's mu bias at (8, 2048, direct): -3.2e-3 beside weight gradients of 155) its own value is no scale
    for a rounding error -- the terms' magnitude is, which the weight gradient shows: the scale of a bias gradient is at least
    1e-2 of its weight gradient's."""
    w = k[:-4] + "weight"
    if not k.endswith(".bias") or grads.get(w) is None:
        return 0.0
    g = grads[w]
    return 1e-2 * float(g.abs().max() if torch.is_tensor(g) else np.abs(g).max())


def close_but_kinks(got, ref, tol, what, KINK_FRACTION=2e-4):
    """rel <= tol everywhere except at a handful of isolated elements: a ReLU whose pre-activation is
    within rounding of zero (|h| ~ 1e-6, there is one in the w12 golden cases) may switch the other way,
    which leaves the outputs continuous but moves that point's gradient to the other subgradient."""
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    err = np.abs(got - ref) / (np.abs(ref).max() + 1e-30)
    bad = int((err > tol).sum())
    assert bad <= max(3, int(KINK_FRACTION * err.size)), (what, bad, float(err.max()))
    assert float(np.median(err)) <= tol / 10, (what, float(np.median(err)))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz")), json.load(open(os.path.join(golden_dir, name + ".json")))


def _check_gproj(named_grads, gold, prefix, seed, tol=1e-3):
    """(sum, random projection, 1-norm) of every parameter gradient against the golden: tests/gradcheck.py"""
    from tests.gradcheck import check_projections
    check_projections(dict(named_grads), _grad_projection(named_grads, seed), lambda k: gold[prefix + "/gproj/" + k], tol, prefix)


@pytest.fixture(params=["bf16x6", "f16x3", "bf16x3"])
def prec(request, monkeypatch):
    """Training precision (of the forward contraction / ReLU masks).  bf16x6, the default, is held to the
    tolerances above with (almost) no ReLU flips; bf16x3 gets the same tolerances on all but <= 1 % of the
    gradient elements and a 3x looser bound on gradients that sum over all points."""
    _gpu()
    from dpf_nets_amd.networks import train_engine
    monkeypatch.setattr(train_engine, "TRAIN_PRECISION", request.param)
    return request.param


def test_single_layer_training_vs_reference_golden(golden_dir, prec):
    nets = _gpu()
    kf = 2e-4 if prec in ("bf16x6", "f16x3") else 1e-2
    gold, meta = _load(golden_dir, "flow_layer")
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    for case in meta["cases"]:
        if case["bn"] != "train":
            continue
        mod = nets.CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
        mod.load_state_dict(FO.to_torch(FO.make_layer_state(case["seed"], F, G, case["warp"])), strict=True)
        mod = mod.cuda().train()
        p, g, r1, r2, r3 = layer_inputs(case["seed"], B, N, G)
        tp = torch.from_numpy(p.copy()).cuda().requires_grad_(True)
        tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
        po, mu, lv = mod(tp, tg, mode=case["mode"])
        t = case["tag"]
        for name, got in (("p_out", po), ("mu", mu), ("logvar", lv)):
            r = rel(got, gold[t + "/" + name])
            assert r <= OUT_REL, (t, name, r)
        loss = (po * torch.from_numpy(r1).cuda()).sum() + (lv * torch.from_numpy(r2).cuda()).sum() \
            + (mu * torch.from_numpy(r3).cuda()).sum()
        loss.backward()
        close_but_kinks(tp.grad, gold[t + "/grad_p"], GRAD_REL[prec], t + " grad_p", kf)
        assert rel(tg.grad, gold[t + "/grad_g"]) <= 2 * GRAD_REL[prec], (t, rel(tg.grad, gold[t + "/grad_g"]))
        _check_gproj([(k, v.grad.cpu()) for k, v in mod.named_parameters()], gold, t, case["seed"], tol=GRAD_REL[prec])
        sd = mod.state_dict()
        for k in sd:
            if k.endswith("running_mean") or k.endswith("running_var"):
                np.testing.assert_allclose(sd[k].cpu().numpy(), gold[t + "/stats/" + k], rtol=1e-4, atol=1e-5, err_msg=t + k)


def test_decoder_training_step_vs_reference_golden(golden_dir, prec):
    """training.py:37-55: inverse flow + PointFlowNLL + backward, n_flows = 2 (6 coupling layers)."""
    nets = _gpu()
    kf, loose = (2e-4, 1.0) if prec in ("bf16x6", "f16x3") else (3e-2, 10.0)   # 384 points: one flipped ReLU moves every sum
    STACK_OUT_REL, STACK_GRAD_REL = STACK_TOL[prec]
    gold, meta = _load(golden_dir, "flow_decoder")
    case = [c for c in meta["cases"] if c.get("bn") == "train"][0]
    c, n_flows, B, N, G, seed = (case[k] for k in ("tag", "n_flows", "B", "N", "G", "seed"))
    dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
    dec.load_state_dict(FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G)), strict=True)
    dec = dec.cuda().train()
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    tp = torch.from_numpy(tgt.copy()).cuda().requires_grad_(True)
    tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
    ps, mus, lvs = dec(tp, tg, mode="inverse")
    assert isinstance(ps, list) and len(ps) == len(mus) == len(lvs) == 3 * n_flows
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    loss = nets.PointFlowNLL()(ps + [tp], [pm] + mus, [pl] + lvs)
    loss.backward()
    assert rel(ps[0], gold[c + "/ps0"]) <= STACK_OUT_REL
    assert rel(sum(lvs), gold[c + "/sum_logvars"]) <= STACK_OUT_REL
    np.testing.assert_allclose(float(loss.detach()), float(gold[c + "/nll"]), rtol=5e-5)
    if prec in ("bf16x6", "f16x3"):
        close_but_kinks(tp.grad, gold[c + "/grad_p"], STACK_GRAD_REL, "grad_p", kf)
        assert rel(tg.grad, gold[c + "/grad_g"]) <= STACK_GRAD_REL, rel(tg.grad, gold[c + "/grad_g"])
        _check_gproj([(k, v.grad.cpu()) for k, v in dec.named_parameters()], gold, c, seed, tol=STACK_GRAD_REL)
    else:
        # bf16x3 (opt-in): with only B*N = 384 points behind every BatchNorm sum, ONE ReLU that switches the
        # other way (pre-activation within the 1e-5 forward error of zero) moves every gradient of the stack;
        # which ReLUs do depends on the last bits.  Gradients are checked where the batch is large enough
        # (test_training_hip_vs_tensor_op_path) -- here only that they exist and are finite.
        assert torch.isfinite(tp.grad).all() and torch.isfinite(tg.grad).all()
        assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in dec.parameters())
    sd = dec.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), gold[c + "/stats/" + k], rtol=1e-4, atol=1e-5, err_msg=k)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 1, k


# (40, 300): more than 32 clouds -- the second round of the per-cloud totals in the backward prologues -- and ragged tiles
@pytest.mark.parametrize("B,N,mode", [(3, 1000, "inverse"), (8, 2048, "direct"), (2, 40, "inverse"), (40, 300, "inverse"), (33, 64, "direct")])
def test_training_hip_vs_tensor_op_path(B, N, mode, prec):
    """Same module, same inputs: HIP kernels vs forward_torch + autograd on the GPU, in float64 (the
    yardstick) and in fp32 (PyTorch-ROCm ops) -- every output, every input gradient, every parameter
    gradient elementwise.  A parameter gradient that is a heavily cancelling sum over all points (a
    bias of the last layers: 16384 terms of both signs, |sum| ~ 1e-3 of the sum of magnitudes) is held
    to a multiple of what the fp32 tensor-op path itself achieves against float64:
    tolerance = max(stated tolerance, 15 x that error) -- the per-point terms come out of the
    dh0 = W1^T dh1 contraction (r04: fp16 hi/lo, 22 bits, for f16x3: R32_FACTOR = 3; bf16 hi/lo, ~1e-5 per term, for the
    opt-in precisions: 15)."""
    nets = _gpu()
    if prec == "bf16x3" and B == 33:
        pytest.skip("2 112 points: the cancelling bias sums (|sum| ~ 1e-3 of the magnitudes) sit below bf16x3's 1e-5 per-term error")
    _hip_vs_tensor_ops(nets, B, N, mode, prec, 31)


def test_training_gradient_errors_over_seeds_are_the_fp32_tensor_ops_own(monkeypatch):
    """the ReLU-flip floors of the test above (KINK_SUM / P, KINK_CLOUD / N, bias_floor) were re-derived in r05 from what ONE
    seed drew.  What they stand for is a mechanism, and this test holds the mechanism instead of a constant: training-mode BatchNorm
    couples all P points, ~10 pre-activations per pass sit within the forward error of zero at ANY seed, and a ReLU that falls the
    other way than in float64 moves a row of d loss / d g by O(1 / N) and a parameter gradient by O(1 / P) of its scale -- on ANY fp32
    evaluation, the reference's own tensor ops included (tests/diag/gg_seeds.py, 24 seeds at this shape: d/dg error x N up to 4.9
    for the HIP stack and up to 6.3 for the fp32 tensor ops; parameter gradients x P up to 476 and 506).  Over eight seeds of weights
    and inputs at (8, 2048, direct), default precision, both paths against float64: the HIP stack's MEDIAN errors are within 2 x the
    fp32 tensor ops' and its WORST within 1.5 x (d/dg) / 5 x (parameters: eight draws of a heavy tail -- seed 33 puts 205 / P on one
    FiLM-net weight of the HIP stack, seed 35 74 / P on the tensor ops'; over 24 seeds both reach ~500 / P unfloored,
    profiles/r06_gg_seeds.txt).  The two paths draw different ReLUs, so the comparison is of distributions, not per seed."""
    nets = _gpu()
    from dpf_nets_amd.networks import train_engine
    monkeypatch.setattr(train_engine, "TRAIN_PRECISION", "f16x3")
    B, N, mode, n_flows, G = 8, 2048, "direct", 2, 128
    rows = []
    for seed in range(32, 40):
        sd = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
        tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
        res = {}
        for impl in ("hip", "torch", "torch64"):
            dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
            dec.load_state_dict(sd, strict=True)
            dec = dec.cuda().train()
            tp, tg = torch.from_numpy(z.copy()).cuda(), torch.from_numpy(g.copy()).cuda()
            if impl == "torch64":
                dec, tp, tg = dec.double(), tp.double(), tg.double()
            tp.requires_grad_(True)
            tg.requires_grad_(True)
            ps, mus, lvs = dec(tp, tg, mode=mode) if impl == "hip" else dec.forward_torch(tp, tg, mode=mode)
            pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
            (nets.PointFlowNLL()([tp] + ps, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()).backward()
            res[impl] = dict(gg=tg.grad, grads={k: v.grad for k, v in dec.named_parameters() if v.grad is not None})
        h, t, t32 = res["hip"], res["torch64"], res["torch"]
        rows.append((rel(h["gg"], t["gg"]) * N, rel(t32["gg"], t["gg"]) * N,
                     max(rel(h["grads"][k], t["grads"][k], bias_floor(t["grads"], k)) for k in t["grads"]) * B * N,
                     max(rel(t32["grads"][k], t["grads"][k], bias_floor(t["grads"], k)) for k in t["grads"]) * B * N))
    a = np.array(rows)
    print("d/dg error x N (hip, fp32 tensor ops), parameter gradient error x P (hip, fp32): max", a.max(0), "median", np.median(a, 0))
    assert a[:, 0].max() <= 1.5 * a[:, 1].max() + 0.5, a[:, :2]
    assert np.median(a[:, 0]) <= 2.0 * np.median(a[:, 1]) + 0.25, a[:, :2]
    assert a[:, 2].max() <= 5.0 * a[:, 3].max() + 8.0, a[:, 2:]
    assert np.median(a[:, 2]) <= 2.0 * np.median(a[:, 3]) + 8.0, a[:, 2:]


def _hip_vs_tensor_ops(nets, B, N, mode, prec, seed):
    kf, loose = (5e-4, 1.0) if prec in ("bf16x6", "f16x3") else (1e-2, 10.0)
    STACK_OUT_REL, STACK_GRAD_REL = STACK_TOL[prec]
    # (r05, ADVICE r04: the floors are capped -- at (2, 40) they had grown to 5e-2 / 0.4 and checked nothing; what the tiny and
    # ragged shapes' kernel forms compute is pinned much tighter by test_training_kernel_forms_agree below, where every form
    # sees the same ReLU masks)
    STACK_GRAD_REL = max(STACK_GRAD_REL, min(KINK / (B * N), KINK_CAP))
    PARAM_REL = max(2 * loose * STACK_TOL[prec][1], min(KINK_SUM / (B * N), KINK_SUM_CAP))
    n_flows, G = 2, 128
    sd = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    src = tgt if mode == "inverse" else z
    res = {}
    for impl in ("hip", "torch", "torch64"):
        dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
        dec.load_state_dict(sd, strict=True)
        dec = dec.cuda().train()
        tp = torch.from_numpy(src.copy()).cuda()
        tg = torch.from_numpy(g.copy()).cuda()
        if impl == "torch64":
            dec, tp, tg = dec.double(), tp.double(), tg.double()
        tp.requires_grad_(True)
        tg.requires_grad_(True)
        if impl == "hip":
            ps, mus, lvs = dec(tp, tg, mode=mode)
        else:
            ps, mus, lvs = dec.forward_torch(tp, tg, mode=mode)
        pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
        smp = ps + [tp] if mode == "inverse" else [tp] + ps
        loss = nets.PointFlowNLL()(smp, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()
        loss.backward()
        res[impl] = dict(ps=[x.detach() for x in ps], mus=[x.detach() for x in mus], lvs=[x.detach() for x in lvs],
                         loss=float(loss), gp=tp.grad, gg=tg.grad,
                         grads={k: v.grad for k, v in dec.named_parameters()},
                         stats={k: v for k, v in dec.state_dict().items() if "running" in k})
    h, t, t32 = res["hip"], res["torch64"], res["torch"]
    for key in ("ps", "mus", "lvs"):
        for i, (a, b) in enumerate(zip(h[key], t[key])):
            assert rel(a, b) <= STACK_OUT_REL or float(b.abs().max()) == 0.0, (key, i, rel(a, b))
    assert abs(h["loss"] - t["loss"]) <= 5e-5 * abs(t["loss"])
    close_but_kinks(h["gp"], t["gp"], STACK_GRAD_REL, "grad_p", kf)
    # a row of grad_g sums over the N points of ONE cloud: a ReLU that falls the other way than in float64 (pre-activation within
    # the forward error of zero -- which ones do changes with every change of the forward rounding) moves it by ~1/N of its scale,
    # not 1/(B N): (8, 2048, direct) met one of 1.06/N when the training forward took the power-of-two W1 scaling
    GG_REL = max(STACK_GRAD_REL, min(KINK_CLOUD / N, KINK_CAP))
    assert rel(h["gg"], t["gg"]) <= loose * GG_REL, rel(h["gg"], t["gg"])
    for k in t["grads"]:
        if t["grads"][k] is None:          # parameter the loss does not reach (direct mode: last layers' mu nets)
            assert h["grads"][k] is None or float(h["grads"][k].abs().max()) == 0.0, k
            continue
        assert h["grads"][k] is not None, k
        fl = bias_floor(t["grads"], k)
        r, r32 = rel(h["grads"][k], t["grads"][k], fl), rel(t32["grads"][k], t["grads"][k], fl)
        if os.environ.get("DPF_TEST_PRINT_GRAD_REL"):
            print("GRADREL", k, r, r32)
        # (the worst case on record, nvp3's mu bias at (8, 2048, direct): 9.1x with the SLP-vectorised build of r02, 10.3x with
        # the scalar build of r03 -- rounding noise of a sum whose terms cancel to 1e-3 of their magnitudes; hence 15x)
        assert r <= max(PARAM_REL, R32_FACTOR[prec] * r32), (k, r, r32)
    for k in t["stats"]:
        np.testing.assert_allclose(h["stats"][k].cpu().numpy(), t["stats"][k].cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("shrink", [2.0 ** -6, 2.0 ** -10])
def test_training_small_w1_vs_float64(shrink, monkeypatch):
    """f16x3 training forward and backward with a SMALL first-hidden weight (W1 times 2^-6 / 2^-10 in every layer: BatchNorm 1
    renormalises, so the function barely changes, but W1's fp16 lo part would drop into the subnormals): the packed forward W1
    carries a power-of-two scale (csrc/flow_common.h w1_pow2_scale; tfold hands D * 2^k and W2' * 2^-k on, tstats_h1 / tbwd1 /
    tbwd2 take the scale back out of their sums), so outputs and gradients stay at the usual bars against float64."""
    nets = _gpu()
    from dpf_nets_amd.networks import train_engine
    monkeypatch.setattr(train_engine, "TRAIN_PRECISION", "f16x3")
    # (16 384 points: at 2 048 a single ReLU that falls the other way than in float64 -- |y| = 9e-7 of its channel's rms, on a point
    # that carries 1.2 % of the channel's gradient -- was all the first version of this test measured: tests/diag/small_w1.py)
    B, N, G, seed = 8, 2048, 128, 37
    sd = FO.to_torch(FO.make_decoder_state(seed, 2, 64, G))
    for k in sd:
        if k.endswith("_sd1.weight"):
            sd[k] = sd[k] * shrink
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    res = {}
    for impl in ("hip", "torch64"):
        dec = nets.LocalCondRNVPDecoder(2, 64, G, weight_std=0.01)
        dec.load_state_dict(sd, strict=True)
        dec = dec.cuda().train()
        tp, tg = torch.from_numpy(tgt.copy()).cuda(), torch.from_numpy(g.copy()).cuda()
        if impl == "torch64":
            dec, tp, tg = dec.double(), tp.double(), tg.double()
        tp.requires_grad_(True)
        tg.requires_grad_(True)
        ps, mus, lvs = dec(tp, tg, mode="inverse") if impl == "hip" else dec.forward_torch(tp, tg, mode="inverse")
        pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
        nets.PointFlowNLL()(ps + [tp], [pm] + mus, [pl] + lvs).backward()
        res[impl] = dict(ps=[x.detach() for x in ps], lvs=[x.detach() for x in lvs], gp=tp.grad, gg=tg.grad,
                         grads={k: v.grad for k, v in dec.named_parameters()})
    h, t = res["hip"], res["torch64"]
    out_rel, grad_rel = 2e-5, STACK_TOL["f16x3"][1]      # outputs: measured 7e-6 (fp32 tensor ops: 3e-6); unscaled W1: 1.6e-4
    grad_rel = max(grad_rel, min(KINK / (B * N), KINK_CAP))
    bad = [(key, i, rel(a, b)) for key in ("ps", "lvs") for i, (a, b) in enumerate(zip(h[key], t[key])) if rel(a, b) > out_rel]
    for k, gt in t["grads"].items():
        if gt is not None:
            r = rel(h["grads"][k], gt, bias_floor(t["grads"], k))
            if r > min(KINK_SUM / (B * N), KINK_SUM_CAP):
                bad.append((k, r))
    assert not bad, bad
    close_but_kinks(h["gp"], t["gp"], grad_rel, "grad_p", 5e-4)
    assert rel(h["gg"], t["gg"]) <= max(grad_rel, min(KINK_CLOUD / N, KINK_CAP)), rel(h["gg"], t["gg"])


def test_training_path_uses_hip_and_repeats(monkeypatch):
    nets = _gpu()
    from dpf_nets_amd.networks import flows
    assert flows.TRAIN_IMPL == "hip"

    def boom(*a, **k):
        raise AssertionError("tensor-op path used on a CUDA training step")
    monkeypatch.setattr(flows.CondRealNVPFlow3D, "forward_torch", boom)
    dec = nets.LocalCondRNVPDecoder(1, 64, 128).cuda().train()
    tgt, z, g = FO.synthetic_inputs(5, 4, 300, 128)
    outs = []
    for _ in range(2):
        dec.zero_grad()
        tp = torch.from_numpy(tgt.copy()).cuda().requires_grad_(True)
        ps, mus, lvs = dec(tp, torch.from_numpy(g).cuda(), mode="inverse")
        (ps[0].square().mean() + sum(lvs).mean()).backward()
        outs.append((ps[0].detach().clone(), tp.grad.clone(), dec.flows[0].nvp1.T_mu_0[3].weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)       # per-wave slots, per-workgroup partials, fixed-order column sums: no atomics


FORMS_SHAPES = [(2, 40, "inverse"), (33, 64, "direct"), (3, 1000, "inverse"), (8, 2048, "direct"), (20, 2048, "inverse")]
FORMS_CODE = """
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from dpf_nets_amd import networks as nets
from oracle import flow_oracle as FO
out = {}
for B, N, mode in %r:
    sd = FO.to_torch(FO.make_decoder_state(31, 2, 64, 128))
    tgt, z, g = FO.synthetic_inputs(31, B, N, 128)
    dec = nets.LocalCondRNVPDecoder(2, 64, 128, weight_std=0.01)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().train()
    tp = torch.from_numpy((tgt if mode == "inverse" else z).copy()).cuda().requires_grad_(True)
    tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
    ps, mus, lvs = dec(tp, tg, mode=mode)
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    smp = ps + [tp] if mode == "inverse" else [tp] + ps
    loss = nets.PointFlowNLL()(smp, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()
    loss.backward()
    key = "%%d_%%d" %% (B, N)
    out[key + "/gp"] = tp.grad.cpu().numpy(); out[key + "/gg"] = tg.grad.cpu().numpy()
    for k, v in dec.named_parameters():
        if v.grad is not None: out[key + "/" + k] = v.grad.cpu().numpy()
from dpf_nets_amd._lib import lib
out["fallbacks"] = np.array([int(lib().dpf_train_colsum_fallbacks())])
np.savez(sys.argv[1], **out)
"""


def test_training_kernel_forms_agree(tmp_path):
    """The backward kernels come in several forms chosen by batch size -- pass 2 with a branch per WAVE (two tiles per wave) or
    per WORKGROUP, pass 1 split by branch or not, pass 1 finished by its per-cloud ticket or by role workgroups of pass 2, the
    column sums of a layer as role workgroups of the next pass 1 or as their own launch.  Every form runs the same forward
    pass, hence the same ReLU masks: their gradients may differ by rounding only.  Small and ragged shapes (where the float64
    comparison above has to allow for a flipped ReLU) are pinned here at 2e-5 of each gradient's scale across all forms, and the
    two column-sum forms bit for bit (same sums in the same order).  Each form needs its own process: the switches are read once."""
    import subprocess
    import sys
    _gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = FORMS_CODE % (root, FORMS_SHAPES)
    forms = {"default": {}, "pair": {"DPF_TRAIN_SPLIT": "0"}, "split": {"DPF_TRAIN_SPLIT": "1"}, "ticket": {"DPF_TRAIN_ROLES": "0"},
             "split+roles": {"DPF_TRAIN_SPLIT": "1", "DPF_TRAIN_ROLES": "1"}, "colsum launch": {"DPF_TRAIN_FUSE_COLSUM": "0"}}
    res = {}
    for name, env in forms.items():
        f = str(tmp_path / (name.replace(" ", "_").replace("+", "_") + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        res[name] = dict(np.load(f))
    base = res["default"]
    for name, got in res.items():
        # the role workgroups were dispatched in front of the workgroups that wait for them: nobody gave up polling.  A workgroup
        # that does give up produces the same bits itself (the comparisons below hold either way): on a GPU this process has to
        # itself the count is 0 -- a handful is tolerated so that a time-sliced box cannot fail a performance assumption as a
        # parity failure; thousands would mean the dispatch-order assumption is gone
        assert int(got["fallbacks"][0]) <= 16, (name, int(got["fallbacks"][0]))
        assert set(got) == set(base), name
        for k in base:
            if k == "fallbacks":
                continue
            if name == "colsum launch":
                assert np.array_equal(got[k], base[k]), (name, k)
                continue
            scale = float(np.abs(base[k]).max())
            scale = max(scale, bias_floor(base, k))
            if scale == 0.0:
                assert float(np.abs(got[k]).max()) == 0.0, (name, k)
                continue
            err = float(np.abs(got[k].astype(np.float64) - base[k]).max()) / scale
            # a cancelling sum (a bias of the later layers) amplifies the forms' different summation orders: those are held to
            # 2e-4 -- still 25 x below what a flipped ReLU would move them by
            assert err <= (2e-4 if k.endswith(".bias") else 2e-5), (name, k, err)


def test_training_graph_replay_is_bitwise_the_eager_call():
    """csrc/graph_cache.h: a stack call seen twice is recorded and replayed as one hipGraph.  Ten identical steps on
    persistent inputs (the allocator settles into handing the same blocks back, so the later ones are replays) must give bit-identical
    outputs and gradients, equal to a run with the replay switched off (DPF_TRAIN_GRAPH=0 in a child process)."""
    import subprocess
    import sys
    nets = _gpu()
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from dpf_nets_amd import networks as nets\n"
        "from oracle import flow_oracle as FO\n"
        "torch.manual_seed(0)\n"
        "dec = nets.LocalCondRNVPDecoder(2, 64, 128).cuda().train(); dec.flatten_parameters()\n"
        "tgt, z, g = FO.synthetic_inputs(7, 4, 700, 128)\n"
        "tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()\n"
        "sig = []\n"
        "for it in range(10):\n"
        "    dec.zero_grad(set_to_none=True)\n"
        "    ps, mus, lvs = dec(tp, tg, mode='inverse')\n"
        "    (ps[0].square().mean() + sum(lvs).mean()).backward()\n"
        "    sig.append((float(ps[0].double().sum()), float(dec.flows[0].nvp1.T_mu_0[3].weight.grad.double().sum()),\n"
        "                float(dec.flows[1].nvp3.T_logvar_0[0].weight.grad.double().abs().sum())))\n"
        "from dpf_nets_amd._lib import lib\n"
        "print(repr((sig, int(lib().dpf_train_graph_replays()))))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = {}
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DPF_TRAIN_GRAPH=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = eval(r.stdout.strip().splitlines()[-1])
    (sig1, replays1), (sig0, replays0) = outs["1"], outs["0"]
    assert len(set(sig1)) == 1, sig1                      # every step the same bits: eager, recorded, replayed
    assert sig1 == sig0
    assert replays0 == 0 and replays1 >= 4, (replays0, replays1)      # the later steps ran as graphs (forward and backward)


def _bench_shape_losses(nets, steps, n_flows, graph_on):
    """Loss and gradient-norm bytes of `steps` optimizer steps at the bench shape, recording / replay on or off."""
    from dpf_nets_amd import synthetic as SY
    from dpf_nets_amd._lib import lib
    prev = lib().dpf_train_graph_set_enabled(1 if graph_on else 0)
    try:
        torch.manual_seed(0)
        dec = nets.LocalCondRNVPDecoder(n_flows, 64, 128).cuda().train()
        store = dec.flatten_parameters()
        opt = nets.Adam(list(dec.parameters()), lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
        tgt, _, g = SY.synthetic_inputs(3, 32, 2048, 128)
        tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
        pm, pl = torch.zeros(32, 3, 2048).cuda(), torch.full((32, 3, 2048), -3.6).cuda()
        nll = nets.PointFlowNLL()
        r0 = int(lib().dpf_train_graph_replays())
        losses, gnorms = [], []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            ps, mus, lvs = dec(tp, tg, mode="inverse")
            loss = nll(ps + [tp], [pm] + mus, [pl] + lvs)
            loss.backward()
            gnorms.append(store.flat_g.double().norm())
            opt.step()
            losses.append(loss.detach())
        out = (torch.stack(losses).cpu().numpy().tobytes(), torch.stack(gnorms).cpu().numpy().tobytes(),
               store.flat_p.detach().clone())
        return out, int(lib().dpf_train_graph_replays()) - r0
    finally:
        lib().dpf_train_graph_set_enabled(prev)


def test_training_replay_equals_eager_at_bench_shape_with_optimizer():
    """at B=32, N=2048, n_flows=21 (63 layers, 8 pass-1 workgroups per cloud, ~750 launches per step) the
    hipGraph replay of the training step must stay on the eager path's trajectory bit for bit: 32 steps of the mirror Adam
    (training.py:54-56), every loss, every gradient norm and the final weights.  (r02's replay drifted from step ~22 on: a
    captured hipMemsetAsync node was not ordered before the first pass-1 kernel; csrc/zero_fill.h.)"""
    nets = _gpu()
    (l1, g1, w1), replays = _bench_shape_losses(nets, 32, 21, True)
    (l0, g0, w0), replays0 = _bench_shape_losses(nets, 32, 21, False)
    assert replays0 == 0 and replays >= 50, (replays0, replays)          # 2 calls per step, all but the first two replayed
    first = next((i for i in range(32) if l1[4 * i:4 * i + 4] != l0[4 * i:4 * i + 4]), None)
    assert first is None, "losses differ from step %d" % first
    assert g1 == g0
    assert torch.equal(w1, w0)


def test_training_f16x3_range_monitor_switches_to_bf16x6(monkeypatch):
    """The training default f16x3 is exact while the post-BN0 activations stay below 2048.  The per-stack monitor
    (max|gamma0| * sqrt(B*N) + max|beta0|, computed on the device every few calls and read without a host sync) must leave a
    healthy stack alone and move one whose gamma0 exploded to bf16x6 for good, with a warning; the step after the switch
    still agrees with the tensor-op path."""
    nets = _gpu()
    import warnings
    from dpf_nets_amd.networks import train_engine
    monkeypatch.setattr(train_engine, "TRAIN_PRECISION", "f16x3")
    monkeypatch.setattr(train_engine, "F16_CHECK_EVERY", 1)
    torch.manual_seed(0)
    dec = nets.LocalCondRNVPDecoder(1, 64, 128).cuda().train()
    dec.flatten_parameters()
    tgt, z, g = FO.synthetic_inputs(77, 4, 600, 128)
    tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    from dpf_nets_amd.networks.flows import stack_spec
    spec = stack_spec(dec, dec.coupling_layers())
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(3):
            dec(tp, tg, mode="inverse")
            torch.cuda.synchronize()
    assert spec.f16_ok
    with torch.no_grad():
        for lyr in dec.coupling_layers():
            lyr.T_mu_0[1].weight.mul_(100.0)                                  # gamma0 x 100: bound = 100 * sqrt(2400) > 2048
    with pytest.warns(UserWarning, match="trains at bf16x6 from now on"):
        for _ in range(3):
            dec(tp, tg, mode="inverse")
            torch.cuda.synchronize()
    assert not spec.f16_ok
    ps, mus, lvs = dec(tp, tg, mode="inverse")
    with torch.no_grad():
        dec2 = nets.LocalCondRNVPDecoder(1, 64, 128).cuda().train()
        dec2.load_state_dict(dec.state_dict())
        rps, rmus, rlvs = dec2.forward_torch(tp, tg, mode="inverse")
    assert rel(ps[0], rps[0]) <= 2e-4


def test_training_loop_with_optimizer_and_eval_switch():
    """A few optimizer steps the way training.py:37-56 drives the decoder (inverse flow + NLL, backward, Adam),
    plus a Chamfer term through nn_distance's backward; then eval() must see the UPDATED weights (the packed
    eval-mode weights are cached per weight version) and agree with the tensor-op path."""
    nets = _gpu()
    from dpf_nets_amd.metrics.StructuralLosses import nn_distance
    torch.manual_seed(0)
    B, N, G = 8, 512, 128
    dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().train()
    tgt, z, g = FO.synthetic_inputs(41, B, N, G)
    tp, tz, tg = (torch.from_numpy(v).cuda() for v in (tgt, z, g))
    opt = torch.optim.Adam(dec.parameters(), lr=2e-3)
    nll = nets.PointFlowNLL()
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    losses = []
    for it in range(12):
        opt.zero_grad()
        ps, mus, lvs = dec(tp, tg, mode="inverse")
        loss = nll(ps + [tp], [pm] + mus, [pl] + lvs)
        out = dec(tz, tg, mode="direct")[0][-1]
        d1, d2 = nn_distance(out.transpose(1, 2).contiguous(), tp.transpose(1, 2).contiguous())
        total = loss / (3 * N) + 10.0 * (d1.mean() + d2.mean())
        total.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.parameters())
        opt.step()
        losses.append(float(total.detach()))
    assert losses[-1] < losses[0], losses
    assert all(int(m.num_batches_tracked) == 24 for m in dec.modules() if isinstance(m, torch.nn.BatchNorm1d))
    dec.eval()
    with torch.no_grad():
        a = dec(tz, tg, mode="direct")[0][-1]
        b = dec.forward_torch(tz, tg, mode="direct")[0][-1]
    assert rel(a, b) <= OUT_REL, rel(a, b)


def test_flat_parameter_store_matches_per_parameter_path():
    """LocalCondRNVPDecoder.flatten_parameters(): same kernels on the same numbers, so an optimizer loop (the
    reference's `optimizer.zero_grad(); loss.backward(); optimizer.step()`, training.py:54-56, incl. the default
    set_to_none) must leave BITWISE the same parameters, BatchNorm statistics and outputs as the path that hands
    every parameter to autograd; the decoder is called twice per step so gradients accumulate."""
    nets = _gpu()
    import copy
    torch.manual_seed(1)
    B, N, G = 6, 700, 128
    ref = nets.LocalCondRNVPDecoder(2, 64, G).cuda().train()
    flat = copy.deepcopy(ref)
    store = flat.flatten_parameters()
    assert flat.flat_store() is store and store.attached()
    names = [k for k, _ in ref.named_parameters()]
    assert names == [k for k, _ in flat.named_parameters()]
    assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), flat.state_dict().values()))
    n_flat = sum(p.numel() for p in flat.parameters())
    assert store.flat_p.numel() >= n_flat                      # + the zero pads of the conditioner block
    tgt, z, g = FO.synthetic_inputs(43, B, N, G)
    tp, tz = torch.from_numpy(tgt).cuda(), torch.from_numpy(z).cuda()
    nll = nets.PointFlowNLL()
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    outs = []
    for dec in (ref, flat):
        opt = torch.optim.Adam(dec.parameters(), lr=1e-3)
        tg = torch.from_numpy(g).cuda().requires_grad_(True)
        for it in range(4):
            if it == 2 and dec is flat:
                store.zero_grad()                               # the one-op variant
            else:
                opt.zero_grad()
            tg.grad = None
            ps, mus, lvs = dec(tp, tg, mode="inverse")
            loss = nll(ps + [tp], [pm] + mus, [pl] + lvs) / (3 * N)
            loss = loss + dec(tz, tg, mode="direct")[0][-1].square().mean()
            loss.backward()
            opt.step()
        outs.append((loss.detach().clone(), tg.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(ref.named_parameters(), flat.named_parameters()):
        assert torch.equal(a.grad, b.grad), k
    assert store.attached() and flat.flows[0].nvp1.T_mu_0[3].weight.grad.data_ptr() >= store.flat_g.data_ptr()
    # a state dict loads in place and keeps the aliasing; .float()/.cuda() would break it and the next step re-flattens
    flat.load_state_dict(ref.state_dict())
    assert store.attached()
    flat._apply(lambda t: t.clone())
    assert not store.attached()
    ps, _, _ = flat(tp, torch.from_numpy(g).cuda(), mode="inverse")
    assert flat.flat_store() is not store and flat.flat_store().attached()
    ps2, _, _ = ref(tp, torch.from_numpy(g).cuda(), mode="inverse")
    assert torch.equal(ps[0], ps2[0])


def test_mirror_adam_on_flat_and_per_parameter_decoders_agree_bitwise():
    """The reference's Adam variant (AMSGrad + coupled weight decay, train_ae.py:63-64) from networks.optimizers on a
    flattened decoder (one op sequence over the flat buffers) and on a per-parameter decoder (multi-tensor ops): the
    same parameters after four HIP training steps, bit for bit."""
    nets = _gpu()
    import copy
    torch.manual_seed(2)
    B, N, G = 4, 300, 128
    ref = nets.LocalCondRNVPDecoder(1, 64, G).cuda().train()
    flat = copy.deepcopy(ref)
    store = flat.flatten_parameters()
    tgt, z, g = FO.synthetic_inputs(47, B, N, G)
    tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    nll = nets.PointFlowNLL()
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    for dec in (ref, flat):
        opt = nets.Adam(dec.parameters(), lr=2e-3, weight_decay=1e-6, betas=(0.9, 0.995), amsgrad=True)
        sched = nets.LRUpdater(10, cycle_length=2, min_lr=1e-4, max_lr=2e-3, beta1=0.9, min_beta2=0.99, max_beta2=0.995)
        for it in range(4):
            sched(opt, 0, it)
            opt.zero_grad()
            ps, mus, lvs = dec(tp, tg, mode="inverse")
            (nll(ps + [tp], [pm] + mus, [pl] + lvs) / (3 * N)).backward()
            opt.step()
        if dec is flat:
            assert len(opt._flat) == 1 and store.attached()
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k


def test_training_layer_sum_of_logvars_reaches_the_nll():
    """The training stack also returns sum(logvars) (one reduction); PointFlowNLL (losses.py:11-15) takes it instead
    of adding the L tensors, and its gradient reaches every layer's log-variances in the backward -- also when a
    layer's logvar is used on its own next to the sum.  Same loss and gradients as the plain python sum."""
    nets = _gpu()
    from dpf_nets_amd.networks.losses import total_logvar
    torch.manual_seed(5)
    B, N, G = 5, 600, 128
    dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().train()
    tgt, z, g = FO.synthetic_inputs(47, B, N, G)
    tp = torch.from_numpy(tgt).cuda()
    nll = nets.PointFlowNLL()
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    res = []
    for tagged in (True, False):
        dec.zero_grad()
        tg = torch.from_numpy(g).cuda().requires_grad_(True)
        ps, mus, lvs = dec(tp, tg, mode="inverse")
        tag = lvs[-1]._dpf_total
        assert tag[1] == len(lvs) == 6 and tag[2].requires_grad
        if tagged:
            assert total_logvar(lvs) is tag[2] and total_logvar([pl] + lvs) is not tag[2]
        else:
            lvs = [v.view_as(v) for v in lvs]                     # fresh tensor objects: no tags, the plain sum
            assert getattr(lvs[-1], "_dpf_total", None) is None
        loss = nll(ps + [tp], [pm] + mus, [pl] + lvs) / (3 * N) + lvs[1].exp().mean()
        loss.backward()
        res.append((loss.detach(), tg.grad.clone(), [p.grad.clone() for p in dec.parameters()]))
    (la, ga, pa), (lb, gb, pb) = res
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    assert rel(ga, gb) <= 1e-4
    worst = max(rel(a, b) for a, b in zip(pa, pb) if float(b.abs().max()) > 0)
    assert worst <= 1e-3, worst


def test_training_stack_error_budget_vs_float64(golden_dir):
    """instead of a flat tolerance, hold the HIP training path to the error the fp32 tensor-op path ITSELF has
    against float64 on the same inputs.  Both are measured against forward_torch in float64 on the reference's own
    6-layer golden case (flow_decoder.json, bn == 'train') and on a full-size batch; the numbers are printed (pytest -s) and
    written to gpurun_out/train_error_budget.json.
      outputs (ps[0], sum of logvars, loss): HIP <= 2 x the fp32 path's error (floor 1e-6: both are at the rounding level of
        fp32 there);
      gradients (d/dp, d/dg, every parameter): HIP <= 3 x the fp32 path's error.  r02 / r03 (bf16 hi/lo gradient operands, 16
        bits): d/dp 8.2e-5 vs 1.8e-5, d/dg 1.7e-5 vs 5.5e-6, worst parameter gradient 1.0e-4 vs 2.0e-5 on the golden case.
        r04 (fp16 hi/lo with power-of-two scaling, 22 bits): 1.2e-5 / 3.8e-6 / 2.1e-5 there (0.7 x, 0.7 x,
        1.05 x the fp32 path's) and 1.5e-2 / 3.3e-4 / 1.2e-3 vs 2.5e-2 / 2.0e-4 / 5.6e-4 at full size (0.6 x, 1.7 x, 2.1 x)
        -- profiles/r04_train_error_budget.json."""
    nets = _gpu()
    import json as _json
    gold, meta = _load(golden_dir, "flow_decoder")
    case = [c for c in meta["cases"] if c.get("bn") == "train"][0]
    report = {}
    for tag, (n_flows, B, N, G, seed) in (("golden_6_layers", tuple(case[k] for k in ("n_flows", "B", "N", "G", "seed"))),
                                          ("full_size_6_layers", (2, 32, 2048, 128, 77))):
        sd = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
        tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
        res = {}
        for impl in ("hip", "torch", "torch64"):
            dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
            dec.load_state_dict(sd, strict=True)
            dec = dec.cuda().train()
            tp, tg = torch.from_numpy(tgt.copy()).cuda(), torch.from_numpy(g.copy()).cuda()
            if impl == "torch64":
                dec, tp, tg = dec.double(), tp.double(), tg.double()
            tp.requires_grad_(True); tg.requires_grad_(True)
            ps, mus, lvs = dec(tp, tg, mode="inverse") if impl == "hip" else dec.forward_torch(tp, tg, mode="inverse")
            pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
            loss = nets.PointFlowNLL()(ps + [tp], [pm] + mus, [pl] + lvs)
            loss.backward()
            res[impl] = dict(ps0=ps[0].detach(), slv=sum(lvs).detach(), loss=float(loss), gp=tp.grad, gg=tg.grad,
                             grads={k: v.grad for k, v in dec.named_parameters()})
        t = res["torch64"]
        rows = {}
        for key in ("ps0", "slv", "gp", "gg"):
            rows[key] = (rel(res["hip"][key], t[key]), rel(res["torch"][key], t[key]))
        rows["loss"] = (abs(res["hip"]["loss"] - t["loss"]) / abs(t["loss"]), abs(res["torch"]["loss"] - t["loss"]) / abs(t["loss"]))
        worst_h = worst_t = 0.0
        for k in t["grads"]:
            worst_h = max(worst_h, rel(res["hip"]["grads"][k], t["grads"][k]))
            worst_t = max(worst_t, rel(res["torch"]["grads"][k], t["grads"][k]))
        rows["param_grads_worst"] = (worst_h, worst_t)
        report[tag] = {k: {"hip_vs_f64": a, "fp32_tensor_ops_vs_f64": b} for k, (a, b) in rows.items()}
        print(tag, {k: ("%.2e" % a, "%.2e" % b) for k, (a, b) in rows.items()})
        for key in ("ps0", "slv", "loss"):
            a, b = rows[key]
            assert a <= max(2 * b, 1e-6), (tag, key, a, b)
        for key in ("gp", "gg", "param_grads_worst"):
            a, b = rows[key]
            assert a <= 3 * b, (tag, key, a, b)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    _json.dump(report, open(os.path.join(out, "train_error_budget.json"), "w"), indent=1)


@pytest.mark.parametrize("mode", ["direct", "inverse"])
def test_c_abi_backward_block_form_equals_the_pointer_list_form(mode):
    """include/dpf_hip.h: dpf_flow_train_backward takes the three gradients as (L,B,3,N) blocks, dpf_flow_train_backward_lists
    as one pointer per layer and list (what the autograd node uses).  Same kernels, same order: every output bit for bit --
    with all three lists given, and with g_mus / g_lvs absent (NULL = zero)."""
    nets = _gpu()
    import ctypes
    from dpf_nets_amd._lib import lib, check, current_stream
    from dpf_nets_amd.networks import train_engine as TE
    B, N, G, n_flows = 3, 700, 64, 2
    dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.02).cuda().train()
    spec = TE.StackSpec(dec.coupling_layers())
    L, prec = spec.L, TE.PREC["bf16x6"]
    tgt, z, g = FO.synthetic_inputs(5, B, N, G)
    p = torch.from_numpy((tgt if mode == "inverse" else z).copy()).cuda()
    tg = torch.from_numpy(g.copy()).cuda()
    cp, fp = spec.canon_params(), spec.film_params()
    zeros = spec.zeros_on(p.device)
    K = 4 * L
    with torch.no_grad():
        tcanon = torch.cat([cp[i].reshape(-1) if kind == "p" else zeros[:i] for kind, i in spec.cat_plan]).view(L, -1)
        W0 = torch.cat([t.reshape(-1) for t in fp[0::5]]).view(K, 64, G)
        gam, bet = torch.cat(fp[1::5]).view(K, 1, 64), torch.cat(fp[2::5]).view(K, 1, 64)
        W1, b1 = torch.cat([t.reshape(-1) for t in fp[3::5]]).view(K, 64, 64), torch.cat(fp[4::5]).view(K, 1, 64)
        outs, saved = TE._forward_core(p, tg, spec, mode, prec, tcanon, W0, gam, bet, W1, b1)
    p_, g_, tcanon, packed, film, stats, ps, *_rest = saved
    mus, lvs = saved[15], saved[16]
    gen = torch.Generator(device="cuda").manual_seed(11)
    g_ps, g_mus, g_lvs = (torch.randn(ps.shape, device="cuda", generator=gen) * s for s in (1.0, 0.3, 0.1))
    L_ = lib()
    assert L_.dpf_flow_train_canon_floats() == tcanon.shape[1] == 2 * 4484          # the size query of the (L, canon) block

    def run(lists, with_ml):
        ws = torch.empty(L_.dpf_flow_train_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
        dp, tmp = torch.empty_like(p), torch.empty_like(p)
        dcanon, dfm = torch.zeros_like(tcanon), torch.zeros((L, 2, 2, B, 64), device="cuda")
        common = (L, B, N, TE.MODE[mode], prec, spec.meta_host, tcanon.data_ptr(), packed.data_ptr(), film.data_ptr(),
                  stats.data_ptr(), p.data_ptr(), ps.data_ptr(), mus.data_ptr(), lvs.data_ptr())
        tail = (dp.data_ptr(), tmp.data_ptr(), dcanon.data_ptr(), dfm.data_ptr(), spec.eps, ws.data_ptr(), current_stream())
        if lists:
            tab = lambda t: (ctypes.c_void_p * L)(*[t[i].data_ptr() for i in range(L)])
            check(L_.dpf_flow_train_backward_lists(*common, tab(g_ps), tab(g_mus) if with_ml else None,
                                                   tab(g_lvs) if with_ml else None, *tail), "lists")
        else:
            check(L_.dpf_flow_train_backward(*common, g_ps.data_ptr(), g_mus.data_ptr() if with_ml else None,
                                             g_lvs.data_ptr() if with_ml else None, *tail), "blocks")
        torch.cuda.synchronize()
        return dp, dcanon, dfm

    for with_ml in (True, False):
        a, b = run(True, with_ml), run(False, with_ml)
        for x, y, name in zip(a, b, ("dp_in", "dcanon", "dfm")):
            assert torch.isfinite(x).all() and float(x.abs().max()) > 0, name
            assert torch.equal(x, y), (name, with_ml)


def test_training_more_moment_rows_than_one_round_of_the_bn0_prologue():
    """B * N / 256 > 2048 rows of input-moment partials: `bn0_loads` takes its first 2048 rows in registers and the rest in a
    loop (r03 rewrote both).  One triple at B = 36, N = 16 384 (2 304 rows) against the tensor-op path in fp32 on the same
    GPU: outputs, batch statistics and the input gradient."""
    nets = _gpu()
    B, N, G, seed = 36, 16384, 128, 77
    sd = FO.to_torch(FO.make_decoder_state(seed, 1, 64, G))
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    res = {}
    for impl in ("hip", "torch"):
        dec = nets.LocalCondRNVPDecoder(1, 64, G, weight_std=0.01)
        dec.load_state_dict(sd, strict=True)
        dec = dec.cuda().train()
        tp = torch.from_numpy(tgt.copy()).cuda().requires_grad_(True)
        tg = torch.from_numpy(g.copy()).cuda()
        ps, mus, lvs = dec(tp, tg, mode="inverse") if impl == "hip" else dec.forward_torch(tp, tg, mode="inverse")
        loss = (ps[0] * ps[0]).mean() + sum(lvs).mean() + 0.1 * (ps[1] * mus[2]).mean()
        loss.backward()
        res[impl] = dict(ps=[x.detach() for x in ps], lvs=[x.detach() for x in lvs], gp=tp.grad.detach(), loss=float(loss),
                         stats={k: v.detach().clone() for k, v in dec.state_dict().items() if "running" in k})
        del dec, tp, tg, ps, mus, lvs, loss
        torch.cuda.empty_cache()
    h, t = res["hip"], res["torch"]
    for a, b in zip(h["ps"] + h["lvs"], t["ps"] + t["lvs"]):
        assert rel(a, b) <= 2e-4, rel(a, b)
    assert abs(h["loss"] - t["loss"]) <= 1e-4 * abs(t["loss"])
    close_but_kinks(h["gp"], t["gp"], 2e-3, "grad_p", 2e-4)
    for k in t["stats"]:
        np.testing.assert_allclose(h["stats"][k].cpu().numpy(), t["stats"][k].cpu().numpy(), rtol=2e-4, atol=1e-5, err_msg=k)


def test_steady_state_training_step_copies_nothing_from_the_host():
    """r03: `sv[:, :, (0, 2)]` in the running-statistics update was ADVANCED indexing -- an index tensor built on the host and copied
    to the device from pageable memory every step, which blocks the host until the stream has drained (0.6 ms per step, invisible
    in every kernel trace).  A steady-state step of the flattened decoder must issue no host-to-device copy and no `aten::index`,
    `aten::item` or `_local_scalar_dense` at all."""
    nets = _gpu()
    from torch.profiler import profile, ProfilerActivity
    B, N, G = 8, 1024, 128
    dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().train()
    dec.flatten_parameters()
    opt = nets.Adam(list(dec.parameters()), lr=1e-4, amsgrad=True)
    tgt, _, g = FO.synthetic_inputs(3, B, N, G)
    tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
    nll = nets.PointFlowNLL()

    def step():
        opt.zero_grad(set_to_none=True)
        ps, mus, lvs = dec(tp, tg, mode="inverse")
        nll(ps + [tp], [pm] + mus, [pl] + lvs).backward()
        opt.step()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    keys = {e.key: e.count for e in prof.key_averages()}
    bad = {k: c for k, c in keys.items() if "HtoD" in k or "Host -> Device" in k or k in ("aten::index", "aten::item", "aten::_local_scalar_dense", "aten::nonzero")}
    assert not bad, bad
