"""Parity of the fused latent-prior-flow kernel (csrc/gprior.hip through the C ABI: the whole eval-mode
GlobalRNVPDecoder stack in one launch) against

  * the golden vectors captured from the reference's GlobalRNVPDecoder (decoders.py:7-38, flows.py:163-243) by
    oracle/gen_golden.py -- both modes, the generation configs' 7 x 128 on G=128, G=512, a toy and a single row,
  * the CPU restatement oracle/gprior_oracle.py on seeded inputs at batch sizes with one and two rows per
    workgroup and a ragged last workgroup,
  * size-independent properties: inverse(direct(g)) = g, sum_lv = sum of the logvar list, zeros on kept coordinates.

Tolerance: <= 1e-4 relative to the tensor's scale (north star for fp32 paths); the kernel is plain fp32 FMAs in k
order and measures ~1e-6."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle import gprior_oracle as GO

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


def rel(got, ref):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def _canon(state, n_flows, G):
    """The canonical block of include/dpf_hip.h from a reference-named state dict."""
    pieces = []
    for prefix, warp, keep in GO.step_plan(n_flows, G):
        for br in ("mu", "logvar"):
            base = "%sT_%s_0.%s_" % (prefix, br, br)
            pieces += [state[base + k].ravel() for k in ("mlp0.weight", "mlp0_bn.weight", "mlp0_bn.bias", "mlp0_bn.running_mean",
                                                         "mlp0_bn.running_var", "mlp1.weight", "mlp1.bias")]
    return np.concatenate(pieces).astype(np.float32)


def _codes(n_flows):
    return [2 * (i % 2) + k for i in range(n_flows) for k in range(2)]


def test_c_abi_vs_reference_golden(golden_dir):
    _gpu()
    from dpf_nets_amd._lib import lib, check, current_stream
    L = lib()
    gold = np.load(os.path.join(golden_dir, "gprior.npz"))
    meta = json.load(open(os.path.join(golden_dir, "gprior.json")))
    for case, (seed, n_flows, nf, G, B) in meta["cases"].items():
        S = 2 * n_flows
        canon = torch.from_numpy(_canon(GO.make_gprior_state(seed, n_flows, nf, G), n_flows, G)).cuda()
        assert canon.numel() == S * L.dpf_gprior_canon_floats(G, nf)
        packed = torch.empty(L.dpf_gprior_packed_floats(S, G, nf), dtype=torch.float32, device="cuda")
        check(L.dpf_gprior_pack(S, G, nf, 1e-5, canon.data_ptr(), packed.data_ptr(), current_stream()), "pack")
        g = torch.from_numpy(GO.gprior_inputs(seed, B, G)).cuda()
        codes = (ctypes.c_int * S)(*_codes(n_flows))
        for mi, mode in enumerate(("direct", "inverse")):
            gs, mus, lvs = (torch.full((S, B, G), float("nan"), device="cuda") for _ in range(3))
            tot, gout = torch.full((B, G), float("nan"), device="cuda"), torch.full((B, G), float("nan"), device="cuda")
            check(L.dpf_gprior_forward(S, B, G, nf, mi, codes, packed.data_ptr(), g.data_ptr(), gs.data_ptr(), mus.data_ptr(),
                                       lvs.data_ptr(), tot.data_ptr(), gout.data_ptr(), GO.EPS, current_stream()), "forward")
            tag = "%s_eval_%s_" % (case, mode)
            assert rel(gs, gold[tag + "gs"]) <= TOL and rel(mus, gold[tag + "mus"]) <= TOL and rel(lvs, gold[tag + "lvs"]) <= TOL
            assert rel(tot, gold[tag + "lvs"].sum(0)) <= TOL
            assert torch.equal(gout, gs[S - 1] if mode == "direct" else gs[0])
            # exact zeros where the reference's lists have them (the kept coordinates of each step)
            assert np.array_equal(mus.cpu().numpy() == 0, gold[tag + "mus"] == 0)
            # every output is optional
            gout2 = torch.empty_like(gout)
            check(L.dpf_gprior_forward(S, B, G, nf, mi, codes, packed.data_ptr(), g.data_ptr(), None, None, None, None,
                                       gout2.data_ptr(), GO.EPS, current_stream()), "forward")
            assert torch.equal(gout, gout2)
    # argument errors come back as codes, nothing is launched
    assert L.dpf_gprior_forward(2, 1, 7, 8, 0, codes, packed.data_ptr(), g.data_ptr(), None, None, None, None, None, 1e-6, None) != 0
    assert L.dpf_gprior_forward(2, 1, 8, 8, 2, codes, packed.data_ptr(), g.data_ptr(), None, None, None, None, None, 1e-6, None) != 0
    bad = (ctypes.c_int * 2)(0, 5)
    assert L.dpf_gprior_forward(2, 1, 8, 8, 0, bad, packed.data_ptr(), g.data_ptr(), None, None, None, None, None, 1e-6, None) != 0
    assert L.dpf_gprior_forward(2, 0, 8, 8, 0, codes, packed.data_ptr(), None, None, None, None, None, None, 1e-6, None) == 0


def _decoder(nets, seed, n_flows, nf, G):
    dec = nets.GlobalRNVPDecoder(n_flows, nf, G)
    dec.load_state_dict(FO.to_torch(GO.make_gprior_state(seed, n_flows, nf, G)), strict=True)
    return dec.cuda().eval()


@pytest.mark.parametrize("n_flows,nf,G,B", [(7, 128, 128, 64), (7, 128, 512, 50), (3, 40, 24, 301), (2, 256, 64, 513), (1, 8, 2, 3)])
def test_module_vs_oracle_and_round_trip(n_flows, nf, G, B):
    """GlobalRNVPDecoder.forward in eval mode is the HIP launch (FlowList views of its buffers); against the oracle
    on the same seeded weights and inputs, and the tensor-op path on the GPU; direct then inverse returns the input."""
    nets = _gpu()
    seed = 300 + G + B
    dec = _decoder(nets, seed, n_flows, nf, G)
    g = GO.gprior_inputs(seed, B, G)
    tg = torch.from_numpy(g).cuda()
    st = FO.to_torch(GO.make_gprior_state(seed, n_flows, nf, G))
    from dpf_nets_amd.networks.flowlist import FlowList
    from dpf_nets_amd.networks.losses import total_logvar
    for mode in ("direct", "inverse"):
        gs, mus, lvs = dec(tg, mode=mode)
        assert isinstance(gs, FlowList) and len(gs) == len(mus) == len(lvs) == 2 * n_flows
        ref = GO.global_rnvp_decoder(st, n_flows, torch.from_numpy(g), mode)
        with torch.no_grad():
            tor = dec.forward_torch(tg, mode)
        for got, r, t in zip((gs, mus, lvs), ref, tor):
            assert rel(got.stacked, torch.stack(r)) <= TOL
            assert rel(got.stacked, torch.stack(t)) <= TOL
        assert rel(total_logvar(lvs), torch.stack(ref[2]).sum(0)) <= TOL
        assert rel(total_logvar([tg] + lvs), torch.stack(ref[2]).sum(0) + torch.from_numpy(g)) <= TOL   # models.py:137-141
    z = dec(tg, mode="direct")[0][-1]
    back = dec(z, mode="inverse")[0][0]
    assert rel(back, g) <= 1e-4
    # the callers' list arithmetic (models.py:139-141, 222-225)
    gs, mus, lvs = dec(tg, mode="inverse")
    samples = gs + [tg]
    assert len(samples) == 2 * n_flows + 1 and samples[-1] is tg
    acc = [tg]
    acc += mus
    assert len(acc) == 2 * n_flows + 1


def test_couple_single_flow_and_other_index_sets():
    """RealNVPFlowCouple / RealNVPFlow called on their own take the same kernel; a RealNVPFlow with warp_inds outside the
    couples' patterns (the constructor default [0]) runs as tensor ops."""
    nets = _gpu()
    torch.manual_seed(3)
    G, nf, B = 32, 24, 7
    g = torch.randn(B, G, device="cuda")
    for pattern in (0, 1):
        cp = nets.RealNVPFlowCouple(nf, G, weight_std=0.1, pattern=pattern).cuda().eval()
        for m in cp.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
        for mode in ("direct", "inverse"):
            got = cp(g, mode=mode)
            assert cp.__dict__.get("_stack") is not None
            with torch.no_grad():
                ref = cp.forward_torch(g, mode)
            for a, b in zip(got, ref):
                assert rel(torch.stack(a), torch.stack(b)) <= TOL
            one = cp.nvp2(g, mode=mode)
            with torch.no_grad():
                ref1 = cp.nvp2.forward_torch(g, mode)
            assert cp.nvp2._stack is not None and all(rel(a, b) <= TOL for a, b in zip(one, ref1))
    odd = nets.RealNVPFlow(nf, G, weight_std=0.1).cuda().eval()          # warp_inds=[0]
    out = odd(g, mode="direct")
    assert odd._stack is None and out[0].shape == g.shape and (out[1][:, 1:] == 0).all()


def test_repack_on_weight_change_training_mode_and_loud_failure():
    nets = _gpu()
    dec = _decoder(nets, 9, 2, 16, 12)
    g = torch.from_numpy(GO.gprior_inputs(9, 6, 12)).cuda()
    a = dec(g)[0][-1].clone()
    with torch.no_grad():
        dec.flows[0].nvp1.T_mu_0[0].weight.mul_(1.5)                    # bumps the sentinel's version counter
        dec.flows[1].nvp1.T_mu_0[3].bias.add_(0.25)
    b = dec(g)[0][-1]
    with torch.no_grad():
        ref = dec.forward_torch(g)[0][-1]
    assert not torch.equal(a, b) and rel(b, ref) <= TOL
    # autograd on the input or train() -> tensor ops (python lists, attached to the graph)
    gr = g.clone().requires_grad_(True)
    gs, _, _ = dec(gr)
    assert isinstance(gs, list) and gs[-1].requires_grad
    dec.train()
    assert isinstance(dec(g)[0], list)
    dec.eval()
    with pytest.raises(RuntimeError, match="MI355X only"):
        dec.stack().run(g.cpu(), "direct")
    assert len(dec(g[:0])[0]) == 4 and dec(g[:0])[0][0].shape == (0, 12)


def _grad_projection(named_grads, seed):
    from oracle.gen_golden import _grad_projection as gp
    return gp(named_grads, seed)


def test_training_mode_vs_reference_golden(golden_dir):
    """GlobalRNVPDecoder under train() on CUDA tensors runs csrc/gprior_train.hip (forward with the statistics of the B
    rows + the whole backward, one autograd node): the three lists, d/dg, every parameter gradient (projections) and the
    BatchNorm running statistics against the vectors captured from the reference's module."""
    nets = _gpu()
    from oracle import detrng
    from dpf_nets_amd.networks import prior_flows as PF
    gold = np.load(os.path.join(golden_dir, "gprior.npz"))
    meta = json.load(open(os.path.join(golden_dir, "gprior.json")))
    for case, (seed, n_flows, nf, G, B) in meta["cases"].items():
        if B < 2:
            continue
        for mode in ("direct", "inverse"):
            tag = "%s_train_%s_" % (case, mode)
            dec = nets.GlobalRNVPDecoder(n_flows, nf, G)
            dec.load_state_dict(FO.to_torch(GO.make_gprior_state(seed, n_flows, nf, G)), strict=True)
            dec = dec.cuda().train()
            g = torch.from_numpy(GO.gprior_inputs(seed, B, G)).cuda().requires_grad_(True)
            calls = []
            orig = PF._GPriorTrain.apply
            PF._GPriorTrain.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
            try:
                lists = dec(g, mode=mode)
            finally:
                PF._GPriorTrain.apply = orig
            assert calls, "the HIP training path was not taken"
            loss = 0.0
            for name, lst in zip(("gs", "mus", "lvs"), lists):
                assert rel(torch.stack(lst), gold[tag + name]) <= TOL, (case, mode, name)
                r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "gprior_r_" + name), (len(lst), B, G))).cuda()
                loss = loss + (torch.stack(lst) * r).sum()
            loss.backward()
            assert rel(g.grad, gold[tag + "dg"]) <= 1e-3, (case, mode, rel(g.grad, gold[tag + "dg"]))
            for k, v in _grad_projection([(k, p.grad.cpu()) for k, p in dec.named_parameters()], seed).items():
                ref = gold[tag + "gproj_" + k]
                np.testing.assert_allclose(v, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref[2])), err_msg=case + mode + k)
            for k, v in dec.state_dict().items():
                if "running" in k:
                    assert rel(v, gold[tag + "stat_" + k]) <= 1e-5, k
                if "num_batches" in k:
                    assert int(v) == 1


def test_training_c_abi_rejects_bad_arguments():
    _gpu()
    from dpf_nets_amd._lib import lib
    L = lib()
    codes = (ctypes.c_int * 2)(0, 1)
    t = torch.zeros(4096, device="cuda")
    p = t.data_ptr()
    assert L.dpf_gprior_train_workspace_floats(4, 8, 16) == 4 * (3 * 32 + 24)
    # a single row has no batch statistics; odd G; unknown step code; missing buffers
    assert L.dpf_gprior_train_forward(2, 1, 8, 16, 0, codes, 0, p, p, p, p, p, p, p, p, 1e-5, 1e-6, None) == -1
    assert L.dpf_gprior_train_forward(2, 4, 7, 16, 0, codes, 0, p, p, p, p, p, p, p, p, 1e-5, 1e-6, None) == -1
    assert L.dpf_gprior_train_forward(2, 4, 8, 16, 0, (ctypes.c_int * 2)(0, 4), 0, p, p, p, p, p, p, p, p, 1e-5, 1e-6, None) == -1
    assert L.dpf_gprior_train_forward(2, 4, 8, 16, 0, codes, 1, p, p, p, p, p, None, p, p, 1e-5, 1e-6, None) == -1
    assert L.dpf_gprior_train_backward(2, 4, 8, 16, 0, codes, 0, p, p, p, p, p, p, p, None, None, None, None, p, p, 1e-5, 1e-6, None) == -1


def test_training_mode_vs_tensor_ops_at_size_and_partial_use():
    """B=64, G=512 and B=50, G=128 against the tensor-op path on the GPU (same weights, same inputs); a loss that uses
    only some of the outputs (None gradients for the rest); two steps accumulate gradients."""
    nets = _gpu()
    import copy
    for (n_flows, nf, G, B, mode) in ((7, 128, 512, 64, "inverse"), (7, 128, 128, 50, "direct"), (2, 24, 20, 130, "inverse")):
        torch.manual_seed(G)
        hipd = nets.GlobalRNVPDecoder(n_flows, nf, G, weight_std=0.05).cuda().train()
        tord = copy.deepcopy(hipd)
        g0 = torch.randn(B, G, device="cuda")
        outs = []
        for dec, use_torch in ((hipd, False), (tord, True)):
            for it in range(2):
                g = g0.clone().requires_grad_(True)
                gs, mus, lvs = dec.forward_torch(g, mode) if use_torch else dec(g, mode=mode)
                first = gs[0] if mode == "inverse" else gs[-1]
                loss = first.square().mean() + sum(lvs).mean() + (mus[1] * 0.5).sum() * 1e-3
                loss.backward()
            outs.append((first.detach(), g.grad, [p.grad for p in dec.parameters()], [b for k, b in dec.state_dict().items() if "running" in k]))
        (fa, ga, pa, sa), (fb, gb, pb, sb) = outs
        assert rel(fa, fb) <= TOL and rel(ga, gb) <= 1e-3
        worst = max(rel(a, b) for a, b in zip(pa, pb) if float(b.abs().max()) > 0)
        assert worst <= 2e-3, worst
        assert max(rel(a, b) for a, b in zip(sa, sb)) <= 1e-5


def test_eval_after_optimizer_steps_uses_the_new_weights():
    """The reference's loop (training.py:54-56 then evaluate): optimizers write through `.data`, which no version counter
    sees -- the train()/eval() switch is what invalidates the packed weights."""
    nets = _gpu()
    dec = _decoder(nets, 11, 2, 16, 12)
    g = torch.from_numpy(GO.gprior_inputs(11, 6, 12)).cuda()
    before = dec(g)[0][-1].clone()
    opt = nets.Adam(dec.parameters(), lr=5e-2, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=True)
    dec.train()
    for _ in range(3):
        opt.zero_grad()
        gs, mus, lvs = dec(g, mode="inverse")
        (gs[0].square().mean() + sum(lvs).mean()).backward()
        opt.step()
    dec.eval()
    after = dec(g)[0][-1]
    with torch.no_grad():
        ref = dec.forward_torch(g)[0][-1]
    assert rel(after, ref) <= TOL and not torch.allclose(before, after)


def test_flat_parameter_store_matches_per_parameter_path_bitwise():
    """GlobalRNVPDecoder.flatten_parameters(): the same kernels on the same numbers, so the reference's loop
    (`optimizer.zero_grad(); loss.backward(); optimizer.step()`, training.py:54-56) must leave BITWISE the same parameters,
    gradients, optimizer state and running statistics as the path that hands every parameter to autograd."""
    nets = _gpu()
    import copy
    torch.manual_seed(2)
    n_flows, nf, G, B = 3, 32, 16, 9
    ref = nets.GlobalRNVPDecoder(n_flows, nf, G, weight_std=0.05).cuda().train()
    flat = copy.deepcopy(ref)
    store = flat.flatten_parameters()
    assert flat.flat_store() is store and store.attached() and store.flat_p.numel() == sum(p.numel() for p in flat.parameters())
    assert [k for k, _ in ref.named_parameters()] == [k for k, _ in flat.named_parameters()]
    g = torch.randn(B, G, device="cuda")
    res = []
    for dec in (ref, flat):
        opt = nets.Adam(dec.parameters(), lr=1e-2, weight_decay=1e-6, betas=(0.9, 0.995), amsgrad=True)
        gin = g.clone().requires_grad_(True)
        for it in range(4):
            if it == 2 and dec is flat:
                store.zero_grad()
            else:
                opt.zero_grad()
            gin.grad = None
            gs, mus, lvs = dec(gin, mode="inverse")
            loss = gs[0].square().mean() + sum(lvs).mean() + dec(gin, mode="direct")[0][-1].abs().mean()     # two uses: gradients add up
            loss.backward()
            opt.step()
        res.append((loss.detach().clone(), gin.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for (k, a), b in zip(ref.state_dict().items(), flat.state_dict().values()):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(ref.named_parameters(), flat.named_parameters()):
        assert torch.equal(a.grad, b.grad), k
    assert store.attached() and flat.flows[0].nvp1.T_mu_0[0].weight.grad.data_ptr() == store.flat_g.data_ptr()
    # eval after training sees the trained weights through the views; .float()/.cuda() break the aliasing, the next step re-flattens
    flat.eval(); ref.eval()
    assert torch.equal(flat(g)[0][-1], ref(g)[0][-1])
    flat._apply(lambda t: t.clone())
    assert flat.flat_store() is None
    flat.train()
    flat(g, mode="inverse")
    assert flat.flat_store() is not None and flat.flat_store() is not store
