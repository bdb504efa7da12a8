"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol
include/dpf_hip.h declares, the module mirror keeps the reference's state-dict contract, the
training-mode tensor-op path matches the reference's golden vectors, FlowList/PointFlowNLL
semantics.  No compute calls into the HIP library here (there is no GPU)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle.gen_golden import layer_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dpf_nets_amd import _lib
    assert _lib.have_lib(), "libdpf_hip.so is not built (run __graft_entry__.build())"
    header = open(os.path.join(ROOT, "include", "dpf_hip.h")).read()
    declared = set(re.findall(r"\b(dpf_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = _lib.lib()                                   # raises if any symbol is unresolved
    assert handle.dpf_version().startswith(b"dpf_hip gfx950")
    for G in (128, 512):                                  # size queries are host-only
        assert handle.dpf_flow_canon_floats(G) == 2 * (4740 + 2 * (64 * G + 4416))
    assert handle.dpf_flow_packed_bytes(14, 128, _lib.PREC["bf16x3"]) == 14 * (2 * 16384 + 4096) + 4 * (14 * 4 * (64 * 128 + 4288) + 14 * 2 * 320)
    assert handle.dpf_flow_film_floats(14, 32) == 14 * 32 * 512


def test_missing_library_fails_loudly(monkeypatch):
    from dpf_nets_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "lib_path", lambda: "/nonexistent/libdpf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_state_dict_contract(golden_dir):
    from dpf_nets_amd.networks import CondRealNVPFlow3D, LocalCondRNVPDecoder, SharedDot
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for warp in ([0], [0, 1]):
        ref = keys["CondRealNVPFlow3D_w" + "".join(map(str, warp))]
        mod = CondRealNVPFlow3D(64, 128, warp_inds=warp)
        sd = mod.state_dict()
        assert [k for k, _, _ in ref] == list(sd.keys())
        assert all(tuple(s) == tuple(sd[k].shape) and d == str(sd[k].dtype) for k, s, d in ref)
        assert keys["CondRealNVPFlow3D_w" + "".join(map(str, warp)) + "_params"] == [k for k, _ in mod.named_parameters()]
    dec = LocalCondRNVPDecoder(2, 64, 512)
    ref = keys["LocalCondRNVPDecoder_nf2_g512"]
    assert [k for k, _, _ in ref] == list(dec.state_dict().keys())
    assert sum(p.numel() for p in dec.parameters()) == keys["LocalCondRNVPDecoder_nf2_g512_nparams"]
    assert [[k, list(v.shape)] for k, v in SharedDot(3, 5, 1, bias=True).state_dict().items()] == keys["SharedDot_3_5_bias"]
    # init statistics (flows.py:52-58, layers.py:29-38)
    torch.manual_seed(0)
    mod = CondRealNVPFlow3D(64, 128, weight_std=0.01, warp_inds=[0])
    st = keys["init_stats"]
    assert abs(float(mod.T_mu_0[0].weight.abs().max()) - st["sd0_absmax"]) < 0.02      # bound sqrt(6/(64*2))
    assert float(mod.T_mu_0[0].weight.abs().max()) <= (6.0 / 128) ** 0.5 + 1e-6
    assert float(mod.T_mu_0[3].weight.abs().max()) <= (6.0 / 4096) ** 0.5 + 1e-6
    assert abs(float(mod.T_mu_1[-1].weight.std()) - 0.01) < 0.004
    assert abs(float(mod.T_mu_0_cond_w[-1].weight.std()) - 0.01) < 0.001
    assert float(mod.T_mu_1[-1].bias.abs().max()) == 0 and float(mod.T_mu_0_cond_b[-1].bias.abs().max()) == 0
    assert float(mod.eps) == pytest.approx(1e-6)


def test_training_path_matches_reference_golden(golden_dir):
    """The tensor-op (training-mode BN) path of the mirror module == the reference module."""
    from dpf_nets_amd.networks import CondRealNVPFlow3D
    gold = np.load(os.path.join(golden_dir, "flow_layer.npz"))
    meta = json.load(open(os.path.join(golden_dir, "flow_layer.json")))
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    for case in meta["cases"]:
        if case["bn"] != "train":
            continue
        mod = CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
        mod.load_state_dict(FO.to_torch(FO.make_layer_state(case["seed"], F, G, case["warp"])), strict=True)
        mod.train()
        p, g, r1, r2, r3 = layer_inputs(case["seed"], B, N, G)
        tp = torch.from_numpy(p.copy()).requires_grad_(True)
        po, mu, lv = mod(tp, torch.from_numpy(g), mode=case["mode"])
        t = case["tag"]
        np.testing.assert_allclose(po.detach().numpy(), gold[t + "/p_out"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(lv.detach().numpy(), gold[t + "/logvar"], rtol=1e-5, atol=2e-6)
        ((po * torch.from_numpy(r1)).sum() + (lv * torch.from_numpy(r2)).sum() + (mu * torch.from_numpy(r3)).sum()).backward()
        np.testing.assert_allclose(tp.grad.numpy(), gold[t + "/grad_p"], rtol=2e-4, atol=2e-5)
        sd = mod.state_dict()
        for k in sd:
            if k.endswith("running_mean") or k.endswith("running_var"):
                np.testing.assert_allclose(sd[k].numpy(), gold[t + "/stats/" + k], rtol=1e-5, atol=1e-6)


def test_eval_path_refuses_cpu():
    from dpf_nets_amd.networks import LocalCondRNVPDecoder
    dec = LocalCondRNVPDecoder(1, 64, 128).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.zeros(1, 3, 8), torch.zeros(1, 128))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        from dpf_nets_amd.metrics.StructuralLosses import nn_distance
        nn_distance(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))


def test_flowlist_and_nll_semantics():
    from dpf_nets_amd.networks.flowlist import FlowList
    from dpf_nets_amd.networks.losses import PointFlowNLL, total_logvar
    torch.manual_seed(0)
    buf = torch.randn(5, 2, 3, 7)
    fl = FlowList(buf, buf.sum(0))
    assert len(fl) == 5 and torch.equal(fl[0], buf[0]) and torch.equal(fl[-1], buf[4])
    with pytest.raises(IndexError):
        fl[5]
    prior = torch.randn(2, 3, 7)
    lst = [prior] + fl                                   # models.py:169-171
    assert isinstance(lst, list) and len(lst) == 6 and lst[1] is fl[0]
    lst2 = [prior]
    lst2 += fl                                           # models.py:119-122
    assert len(lst2) == 6
    assert isinstance(fl + [prior], list)                # models.py:168: buf_p[0] + [p_input]
    np.testing.assert_allclose(total_logvar(lst).numpy(), sum([prior] + list(buf.unbind(0))).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(total_logvar(lst2).numpy(), total_logvar(lst).numpy())
    np.testing.assert_allclose(sum(fl).numpy(), buf.sum(0).numpy(), rtol=1e-6, atol=1e-6)
    # a list that merely ENDS with a foreign tagged view must not use the fused total
    other = FlowList(torch.randn(5, 2, 3, 7), torch.zeros(2, 3, 7))
    mixed = [prior] + list(buf.unbind(0))[:4] + [other[4]]
    np.testing.assert_allclose(total_logvar(mixed).numpy(), sum(mixed).numpy())
    # PointFlowNLL == oracle formula
    smp, mus = [torch.randn(2, 3, 7)], [torch.randn(2, 3, 7)]
    ref = FO.point_flow_nll(smp, mus, lst)
    np.testing.assert_allclose(float(PointFlowNLL()(smp, mus, lst)), float(ref), rtol=1e-6)


def test_synthetic_generator_matches_oracle_streams():
    from dpf_nets_amd import synthetic as SY
    a, b = SY.make_decoder_state(3, 2, 64, 128), FO.make_decoder_state(3, 2, 64, 128)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert all(np.array_equal(x, y) for x, y in zip(SY.synthetic_inputs(1, 2, 16, 8), FO.synthetic_inputs(1, 2, 16, 8)))


def test_canonical_block_layout():
    """engine.layer_canon_pieces follows the layout documented in include/dpf_hip.h."""
    from dpf_nets_amd.networks import CondRealNVPFlow3D
    from dpf_nets_amd.networks.engine import layer_canon_pieces, layer_meta
    from dpf_nets_amd import _lib
    for warp, G in (([1], 128), ([0, 2], 512)):
        mod = CondRealNVPFlow3D(64, G, warp_inds=warp)
        flat = torch.cat(layer_canon_pieces(mod))
        assert flat.numel() == _lib.lib().dpf_flow_canon_floats(G)
        br = flat.numel() // 2
        nk = 3 - len(warp)
        w0 = flat[:128].view(64, 2)                         # logvar branch first
        assert torch.equal(w0[:, :nk], mod.T_logvar_0[0].weight[0]) and (w0[:, nk:] == 0).all()
        assert torch.equal(flat[384:384 + 4096].view(64, 64), mod.T_logvar_0[3].weight[0])
        w2 = flat[4608:4736].view(2, 64)
        assert torch.equal(w2[:len(warp)], mod.T_logvar_1[1].weight[0]) and (w2[len(warp):] == 0).all()
        assert torch.equal(flat[br + 384:br + 384 + 4096].view(64, 64), mod.T_mu_0[3].weight[0])
        assert torch.equal(flat[4740:4740 + 64 * G].view(64, G), mod.T_logvar_0_cond_w[0].weight)
        m = layer_meta(mod)
        assert m[:2] == ([c for c in range(3) if c not in warp] + [-1])[:2] and m[2:] == (warp + [-1])[:2]


def test_training_stack_spec_layout_matches_c_abi():
    """The (L, 8968) parameter block the training C ABI takes is ONE concatenation of the parameters as stored
    (+ zero pads): check every offset of include/dpf_hip.h against a hand-built block, for both warp patterns."""
    from dpf_nets_amd.networks import LocalCondRNVPDecoder
    from dpf_nets_amd.networks.train_engine import StackSpec
    torch.manual_seed(3)
    dec = LocalCondRNVPDecoder(2, 64, 128)            # pattern 0 (1 warp channel) and pattern 1 (2 warp channels)
    layers = dec.coupling_layers()
    spec = StackSpec(layers)
    cp = spec.canon_params()
    zeros = torch.zeros(64)
    block = torch.cat([cp[i].reshape(-1) if kind == "p" else zeros[:i] for kind, i in spec.cat_plan]).view(spec.L, 8968)
    assert len(spec.all_params()) == 32 * spec.L and spec.all_params()[0] is layers[0].T_logvar_0[0].weight
    for li, lyr in enumerate(layers):
        nk, nw = len(lyr.keep_inds), len(lyr.warp_inds)
        assert spec.metas[li] == tuple(list(lyr.keep_inds) + [-1] * (2 - nk) + list(lyr.warp_inds) + [-1] * (2 - nw))
        for bi, br in enumerate(("logvar", "mu")):
            row = block[li, bi * 4484:(bi + 1) * 4484]
            t0, sd2 = getattr(lyr, "T_%s_0" % br), getattr(lyr, "T_%s_1" % br)[1]
            assert torch.equal(row[0:64 * nk], t0[0].weight.reshape(-1)) and float(row[64 * nk:128].abs().max() if nk < 2 else 0) == 0
            assert torch.equal(row[128:192], t0[1].weight) and torch.equal(row[192:256], t0[1].bias)
            assert torch.equal(row[256:4352], t0[3].weight.reshape(-1))
            assert torch.equal(row[4352:4352 + 64 * nw], sd2.weight.reshape(-1))
            assert torch.equal(row[4480:4480 + nw], sd2.bias.reshape(-1)) and float(row[4480 + nw:4484].abs().max()) == 0
    # gradient slices map back one-to-one
    covered = sum(n for _, n in spec.canon_slots)
    assert covered == sum(p.numel() for p in cp)


def test_flat_store_aliases_parameters_and_survives_zero_grad():
    """train_engine.FlatStore on CPU tensors (no kernels involved): every parameter / BatchNorm buffer becomes a
    view of the flat buffers, keeps its values, name and shape; optimizer.zero_grad(set_to_none=True) is repaired by
    attach_grads(); accumulate() adds the result blocks where the per-parameter path would scatter them."""
    import torch
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.train_engine import StackSpec, FlatStore, _T_BR
    torch.manual_seed(3)
    dec = nets.LocalCondRNVPDecoder(1, 64, 16)
    for p in dec.parameters():
        p.data.normal_()
    before = {k: v.clone() for k, v in dec.state_dict().items()}
    layers = dec.coupling_layers()
    spec = StackSpec(layers)
    fs = spec.flatten(torch.device("cpu"))
    assert isinstance(fs, FlatStore) and spec.flatten(torch.device("cpu")) is fs and fs.attached()
    after = dec.state_dict()
    assert list(before) == list(after) and all(torch.equal(before[k], after[k]) for k in before)
    lo, hi = fs.flat_p.data_ptr(), fs.flat_p.data_ptr() + 4 * fs.flat_p.numel()
    assert all(lo <= p.data_ptr() < hi for p in spec.all_params())
    assert fs.flat_p.numel() == len(layers) * 2 * _T_BR + sum(p.numel() for p in spec.film_params())
    # the conditioner block is exactly what the per-parameter path gathers
    zeros = spec.zeros_on(torch.device("cpu"))
    cp = spec.canon_params()
    gathered = torch.cat([cp[i].reshape(-1) if kind == "p" else zeros[:i] for kind, i in spec.cat_plan])
    assert torch.equal(gathered, fs.blocks[0].reshape(-1))
    # in-place updates go through
    with torch.no_grad():
        layers[0].T_mu_0[3].weight.add_(1.0)
    where = lambda t: next(i for i, q in enumerate(spec.all_params()) if q is t)   # noqa: E731
    assert torch.equal(fs.pviews[where(layers[0].T_mu_0[3].weight)], layers[0].T_mu_0[3].weight)
    # gradients: views of flat_g, repaired after set_to_none, accumulated blockwise
    opt = torch.optim.SGD(dec.parameters(), lr=0.1)
    opt.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in spec.all_params())
    blocks = [torch.randn_like(b) for b in fs.gblocks]
    fs.flat_g.fill_(7.0)                                        # stale content must not leak into the new gradients
    fs.accumulate(blocks[0], blocks[1], blocks[2].squeeze(1), blocks[3].squeeze(1), blocks[4], blocks[5].squeeze(1))
    fs.accumulate(blocks[0], blocks[1], blocks[2].squeeze(1), blocks[3].squeeze(1), blocks[4], blocks[5].squeeze(1))
    assert all(torch.equal(g, 2 * b) for g, b in zip(fs.gblocks, blocks))
    w = layers[0].T_logvar_0_cond_w[3].weight
    k = next(i for i, m in enumerate(spec.film_modules()) if m is layers[0].T_logvar_0_cond_w)
    assert w.grad is not None and torch.equal(w.grad, 2 * blocks[4][k])
    o, n = spec.canon_slots[3]                                  # first branch's W1
    assert torch.equal(cp[3].grad.reshape(-1), 2 * blocks[0].reshape(-1)[o:o + n])
    # a replaced .grad tensor is folded in, not lost
    w.grad = torch.ones_like(w)
    fs.attach_grads(full=True)
    assert w.grad.data_ptr() == fs.gviews[where(w)].data_ptr() and bool((w.grad == 1).all())
    # running statistics live in (8L, F) blocks
    bn = layers[0].T_mu_0[1]
    fs.update_running(torch.ones(fs.nfilm, 64), torch.ones(fs.nfilm, 64), torch.full((fs.nfilm, 64), 2.0),
                      torch.full((fs.nfilm, 64), 3.0), 0.1)
    assert torch.allclose(bn.running_mean, torch.full((64,), 0.2)) and int(bn.num_batches_tracked) == 1
    assert torch.allclose(bn.running_var, torch.full((64,), 0.9 + 0.3))
    opt.step()                                                  # parameters still alias the flat buffer afterwards
    assert fs.attached()


def test_encoder_state_dict_contract(golden_dir):
    """PointNetCloudEncoder: the reference's state-dict keys, shapes (encoders.py:15-25) and init; CPU eval raises."""
    import json
    import os
    import pytest
    import torch
    from dpf_nets_amd import networks as nets
    from oracle import encoder_oracle as EO, flow_oracle as FO
    meta = json.load(open(os.path.join(golden_dir, "encoder.json")))
    enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512])
    assert list(enc.state_dict().keys()) == meta["keys"]
    enc.load_state_dict(FO.to_torch(EO.make_encoder_state(3)), strict=True)
    assert enc.features.sd2.weight.shape == (1, 512, 256) and enc.hip_supported()
    x = torch.from_numpy(EO.encoder_inputs(3, 2, 40))
    enc.train()
    st = FO.to_torch(EO.make_encoder_state(3))
    ref = EO.encoder_features(st, x, training=True)
    assert torch.allclose(enc(x), ref, rtol=1e-4, atol=1e-5)            # the tensor-op path = the reference's ops
    enc.eval()
    with pytest.raises(RuntimeError):
        enc(x)


def test_mirror_adam_flat_store_path_is_bitwise_the_per_parameter_path():
    """networks.optimizers.Adam: the parameters of a FlatStore are updated by ONE sequence of ops over the flat buffers;
    results, per-parameter state entries (views), state_dict round trip and resumption must equal the per-parameter
    path bit for bit (CPU tensors: no kernels involved)."""
    import copy
    import torch
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.train_engine import StackSpec
    torch.manual_seed(5)
    ref = nets.LocalCondRNVPDecoder(1, 64, 16)
    for p in ref.parameters():
        p.data.normal_()
    flat = copy.deepcopy(ref)
    extra_r, extra_f = torch.nn.Parameter(torch.randn(5, 3)), None
    extra_f = torch.nn.Parameter(extra_r.detach().clone())
    fs = StackSpec(flat.coupling_layers()).flatten(torch.device("cpu"))
    kw = dict(lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-3, amsgrad=True)
    o_r = nets.Adam(list(ref.parameters()) + [extra_r], **kw)
    o_f = nets.Adam(list(flat.parameters()) + [extra_f], **kw)
    gen = torch.Generator().manual_seed(1)

    def one_step(opt_r, opt_f):
        grads = [torch.randn(p.shape, generator=gen) for p in ref.parameters()] + [torch.randn(5, 3, generator=gen)]
        opt_f.zero_grad(set_to_none=False)
        for p, g in zip(list(ref.parameters()) + [extra_r], grads):
            p.grad = g.clone()
        fs.attach_grads(full=True)
        for p, g in zip(list(flat.parameters()) + [extra_f], grads):
            p.grad.copy_(g) if p.grad is not None else setattr(p, "grad", g.clone())
        opt_r.step(); opt_f.step()

    for _ in range(3):
        one_step(o_r, o_f)
    assert fs.attached() and len(o_f._flat) == 1                       # the flat path was taken
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k
    assert torch.equal(extra_r, extra_f)
    w = flat.flows[0].nvp2.T_mu_0[3].weight
    st = o_f.state[w]
    lo, hi = o_f._flat[id(fs)]["buf"]["exp_avg"].data_ptr(), o_f._flat[id(fs)]["buf"]["exp_avg"].data_ptr() + 4 * fs.flat_p.numel()
    assert st["step"] == 3 and lo <= st["exp_avg"].data_ptr() < hi
    assert torch.equal(st["max_exp_avg_sq"], o_r.state[ref.flows[0].nvp2.T_mu_0[3].weight]["max_exp_avg_sq"])
    # resume: a fresh optimizer loads the state dict (which copies every tensor) and continues identically
    sd_r, sd_f = copy.deepcopy(o_r.state_dict()), copy.deepcopy(o_f.state_dict())
    n_r = nets.Adam(list(ref.parameters()) + [extra_r], **kw); n_r.load_state_dict(sd_r)
    n_f = nets.Adam(list(flat.parameters()) + [extra_f], **kw); n_f.load_state_dict(sd_f)
    for _ in range(2):
        one_step(n_r, n_f)
    assert len(n_f._flat) == 1
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k
    assert n_f.state[w]["step"] == 5


def test_mirror_adam_skips_a_flat_store_without_a_gradient_like_grad_none():
    """zero_grad() keeps a FlatStore's gradient views attached (zeros), so "this parameter got no gradient" is
    carried by FlatStore.grad_written: after zero_grad(set_to_none=True) a step() with no backward / exchange in between
    must leave weights, moments and step counters of the store alone (what stock PyTorch does for .grad None), while
    set_to_none=False makes the zeros count (weight decay and moment decay apply), as it does there."""
    import torch
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.train_engine import StackSpec
    torch.manual_seed(6)
    dec = nets.LocalCondRNVPDecoder(1, 64, 16)
    for p in dec.parameters():
        p.data.normal_()
    fs = StackSpec(dec.coupling_layers()).flatten(torch.device("cpu"))
    other = torch.nn.Parameter(torch.randn(4))
    opt = nets.Adam(list(dec.parameters()) + [other], lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-2, amsgrad=True)
    fs.flat_g.normal_()
    fs.grad_written = True                                               # as FlatStore.accumulate leaves it
    other.grad = torch.ones(4)
    opt.step()
    w = dec.flows[0].nvp1.T_mu_0[3].weight
    assert opt.state[w]["step"] == 1
    before, m_before = fs.flat_p.clone(), opt._flat[id(fs)]["buf"]["exp_avg"].clone()
    opt.zero_grad()                                                      # set_to_none=True: views stay, flag drops
    assert w.grad is not None and float(w.grad.abs().sum()) == 0.0 and not fs.grad_written and other.grad is None
    opt.step()                                                           # no backward in between: nothing moves
    assert torch.equal(fs.flat_p, before) and torch.equal(opt._flat[id(fs)]["buf"]["exp_avg"], m_before)
    assert opt.state[w]["step"] == 1
    opt.zero_grad(set_to_none=False)                                     # zeros ARE the gradients
    opt.step()
    assert opt.state[w]["step"] == 2 and not torch.equal(fs.flat_p, before)
    # a rebuilt store (module.to() re-assigned .data) invalidates the group cache instead of pinning the slow path
    dec._apply(lambda t: t.clone())
    fs2 = StackSpec(dec.coupling_layers()).flatten(torch.device("cpu"))
    assert fs2 is not fs
    fs2.flat_g.normal_(); fs2.grad_written = True
    opt.step()
    assert id(fs2) in opt._flat                                          # the fast path was taken for the new store


def test_mirror_adam_steps_a_flat_store_whose_gradients_came_by_ordinary_autograd():
    """`grad_written` is a hint.  Gradients that reach flat_g through AccumulateGrad -- the tensor-op
    path of a flattened decoder (forward_torch / DPF_TRAIN_IMPL=torch), a single flow module's own forward -- must make
    the next step() update the store: zero_grad (views stay attached, flag cleared), backward, step -> the parameters moved,
    the step counters advanced.  Both stores (point decoder, latent prior flow)."""
    import torch
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.train_engine import StackSpec
    torch.manual_seed(8)
    dec = nets.LocalCondRNVPDecoder(1, 64, 16)
    fs = StackSpec(dec.coupling_layers()).flatten(torch.device("cpu"))
    opt = nets.Adam(dec.parameters(), lr=1e-2, betas=(0.9, 0.99), amsgrad=True)
    p, g = torch.randn(3, 3, 20), torch.randn(3, 16)
    for it in range(2):
        opt.zero_grad()
        assert not fs.grad_written and dec.flows[0].nvp1.T_mu_0[3].weight.grad is fs.gviews[dec_index(fs, dec.flows[0].nvp1.T_mu_0[3].weight)]
        before = fs.flat_p.clone()
        ps, mus, lvs = dec.forward_torch(p, g, mode="inverse")           # tensor ops: autograd accumulates into the views
        (ps[0].square().mean() + sum(lvs).mean()).backward()
        assert fs.grad_written and float(fs.flat_g.abs().sum()) > 0
        opt.step()
        assert not torch.equal(fs.flat_p, before)
        assert opt.state[dec.flows[0].nvp1.T_mu_0[3].weight]["step"] == it + 1
    # one flow module alone (only ITS parameters receive gradients): the store still steps
    opt.zero_grad()
    before = fs.flat_p.clone()
    out = dec.flows[0].nvp2.forward_torch(p, g, mode="direct") if hasattr(dec.flows[0].nvp2, "forward_torch") else dec.flows[0].nvp2(p, g, mode="direct")
    out[0].square().mean().backward()
    assert fs.grad_written
    opt.step()
    assert not torch.equal(fs.flat_p, before)
    # the latent prior flow's store
    prior = nets.GlobalRNVPDecoder(2, 8, 6, weight_std=0.1)
    store = prior.flatten_parameters()
    o2 = nets.Adam(prior.parameters(), lr=1e-2, amsgrad=True)
    gl = torch.randn(5, 6)
    o2.zero_grad()
    assert not store.grad_written
    before = store.flat_p.clone()
    gs, mus, lvs = prior(gl, mode="inverse")                             # CPU tensors: tensor-op path
    (gs[0].square().mean() + sum(lvs).mean()).backward()
    assert store.grad_written
    o2.step()
    assert not torch.equal(store.flat_p, before)


def dec_index(fs, param):
    return next(i for i, q in enumerate(fs.params) if q is param)


def test_prior_flat_store_aliases_parameters_and_takes_the_flat_adam_path():
    """GlobalRNVPDecoder.flatten_parameters() on CPU tensors (no kernels involved): names, shapes and values unchanged,
    .data / .grad are views of the two flat buffers in the parameters-only canonical layout of include/dpf_hip.h, gradients
    re-attach after optimizer.zero_grad(set_to_none=True), and networks.optimizers.Adam updates the store with one op
    sequence, bit for bit the per-parameter path."""
    import copy
    import torch
    from dpf_nets_amd import networks as nets
    torch.manual_seed(3)
    ref = nets.GlobalRNVPDecoder(2, 8, 6, weight_std=0.1)
    flat = copy.deepcopy(ref)
    before = {k: v.clone() for k, v in flat.state_dict().items()}
    store = flat.flatten_parameters()
    assert flat.flat_store() is store and store.flat_p.numel() == sum(p.numel() for p in flat.parameters()) == 4 * 2 * (2 * 8 * 3 + 2 * 8 + 3)
    assert all(torch.equal(v, before[k]) for k, v in flat.state_dict().items())
    w = flat.flows[0].nvp1.T_mu_0[0].weight                              # first tensor of the layout
    assert w.data_ptr() == store.flat_p.data_ptr() and w.grad.data_ptr() == store.flat_g.data_ptr() and w._dpf_flat is store
    b_last = flat.flows[1].nvp2.T_logvar_0[3].bias                       # and the last one
    assert b_last.data_ptr() == store.flat_p.data_ptr() + 4 * (store.total - 3)
    kw = dict(lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-3, amsgrad=True)
    o_r, o_f = nets.Adam(ref.parameters(), **kw), nets.Adam(flat.parameters(), **kw)
    g = torch.randn(5, 6)
    for it in range(3):
        for dec, opt in ((ref, o_r), (flat, o_f)):
            opt.zero_grad()                                              # set_to_none: the store re-attaches zeroed views
            gs, mus, lvs = dec(g, mode="inverse")                        # CPU tensors: the tensor-op path, autograd per parameter
            (gs[0].square().mean() + sum(lvs).mean()).backward()
            if dec is flat:
                store.attach_grads(full=True)                            # tensor-op gradients are copied into the views
            opt.step()
    assert len(o_f._flat) == 1 and store.attached()
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k
    assert o_f.state[w]["step"] == 3 and torch.equal(o_f.state[w]["exp_avg"], o_r.state[ref.flows[0].nvp1.T_mu_0[0].weight]["exp_avg"])


def test_compiled_structural_losses_extension_builds_and_exports_the_reference_names():
    """pybind/bind.cpp:9-15 of the reference registers five functions; csrc/torch_ext/structural_losses_backend.cpp is the
    same module compiled against libdpf_hip.so's C ABI (no compute here: a CPU tensor is refused with the reference's
    message, structural_loss.cpp:10)."""
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dpf_ext_build", os.path.join(root, "dpf_nets_amd", "csrc", "torch_ext", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend_native as native
    for name in ("ApproxMatch", "MatchCost", "MatchCostGrad", "NNDistance", "NNDistanceGrad"):
        assert callable(getattr(native, name)), name
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        native.NNDistance(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
