"""Pins the CPU oracle (oracle/) against golden vectors captured from the
reference's own PyTorch modules (oracle/gen_golden.py).  CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import detrng
from oracle import flow_oracle as FO
from oracle import structural as S
from oracle.gen_golden import layer_inputs, chamfer_inputs, _grad_projection

RTOL, ATOL = 1e-5, 2e-6   # oracle and reference are both fp32 torch-CPU; only op association differs


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz")), json.load(open(os.path.join(golden_dir, name + ".json")))


def test_param_spec_matches_reference_state_dict(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for warp in ([0], [0, 1]):
        ref = keys["CondRealNVPFlow3D_w" + "".join(map(str, warp))]
        spec = FO.layer_param_spec(64, 128, warp)
        assert [k for k, _, _ in ref] == [k for k, _, _ in spec]
        assert [tuple(s) for _, s, _ in ref] == [tuple(s) for _, s, _ in spec]
    ref = keys["LocalCondRNVPDecoder_nf2_g512"]
    st = FO.make_decoder_state(0, 2, 64, 512)
    assert [k for k, _, _ in ref] == list(st.keys())
    nparam = sum(int(np.prod(s)) for k, s, _ in ref
                 if not (k.endswith("running_mean") or k.endswith("running_var")
                         or k.endswith("num_batches_tracked") or k.endswith("eps")))
    assert nparam == keys["LocalCondRNVPDecoder_nf2_g512_nparams"]
    # 157 058 / 157 060 params per layer at G=512 (pattern 0 / 1)
    assert nparam == 3 * 157058 + 3 * 157060


def test_coupling_layer_vs_reference(golden_dir):
    gold, meta = _load(golden_dir, "flow_layer")
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    for case in meta["cases"]:
        tag, warp, mode, bn, seed = case["tag"], case["warp"], case["mode"], case["bn"], case["seed"]
        st = FO.to_torch(FO.make_layer_state(seed, F, G, warp))
        for k, v in st.items():
            if v.dtype == torch.float32 and k != "eps" and not ("running" in k):
                v.requires_grad_(True)
        p, g, r1, r2, r3 = layer_inputs(seed, B, N, G)
        tp = torch.from_numpy(p.copy()).requires_grad_(True)
        tg = torch.from_numpy(g.copy()).requires_grad_(True)
        stats = {}
        p_out, mu, lv = FO.coupling_layer(st, tp, tg, mode, warp, training=(bn == "train"), stats_out=stats)
        np.testing.assert_allclose(p_out.detach().numpy(), gold[tag + "/p_out"], rtol=RTOL, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(mu.detach().numpy(), gold[tag + "/mu"], rtol=RTOL, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(lv.detach().numpy(), gold[tag + "/logvar"], rtol=RTOL, atol=ATOL, err_msg=tag)
        loss = (p_out * torch.from_numpy(r1)).sum() + (lv * torch.from_numpy(r2)).sum() + (mu * torch.from_numpy(r3)).sum()
        loss.backward()
        np.testing.assert_allclose(tp.grad.numpy(), gold[tag + "/grad_p"], rtol=2e-4, atol=2e-5, err_msg=tag)
        np.testing.assert_allclose(tg.grad.numpy(), gold[tag + "/grad_g"], rtol=2e-3, atol=2e-4, err_msg=tag)
        named = [(k, v.grad) for k, v in st.items() if v.requires_grad]
        for k, v in _grad_projection(named, seed).items():
            ref = gold[tag + "/gproj/" + k]
            scale = ref[2] + 1e-6        # sum |grad|: projection error budget relative to gradient mass
            assert abs(v[0] - ref[0]) <= 2e-4 * scale + 1e-5, (tag, k, v, ref)
            assert abs(v[1] - ref[1]) <= 2e-4 * scale * 3 + 1e-5, (tag, k, v, ref)
        if bn == "train":
            for k, v in stats.items():
                np.testing.assert_allclose(v.numpy(), gold[tag + "/stats/" + k], rtol=1e-5, atol=1e-6, err_msg=tag + k)
            assert len(stats) == 2 * 2 * 4       # 2 branches x (sd0_bn, sd1_bn, film_w0_bn, film_b0_bn) x (mean, var)


def test_decoder_vs_reference(golden_dir):
    gold, meta = _load(golden_dir, "flow_decoder")
    for case in meta["cases"]:
        if case.get("bn") == "train":
            continue
        c, n_flows, B, N, G, seed, mode = (case[k] for k in ("tag", "n_flows", "B", "N", "G", "seed", "mode"))
        st = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
        tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
        src = z if mode == "direct" else tgt
        with torch.no_grad():
            ps, mus, lvs = FO.decoder(st, n_flows, torch.from_numpy(src), torch.from_numpy(g), mode)
            assert len(ps) == 3 * n_flows
            tol = dict(rtol=5e-5, atol=5e-6) if n_flows > 5 else dict(rtol=RTOL, atol=ATOL)
            for k in case["picks"]:
                np.testing.assert_allclose(ps[k].numpy(), gold["%s/ps%d" % (c, k)], err_msg=c, **tol)
                np.testing.assert_allclose(mus[k].numpy(), gold["%s/mus%d" % (c, k)], err_msg=c, **tol)
                np.testing.assert_allclose(lvs[k].numpy(), gold["%s/logvars%d" % (c, k)], err_msg=c, **tol)
            np.testing.assert_allclose(sum(lvs).numpy(), gold[c + "/sum_logvars"], err_msg=c, **tol)
            pm, pl = torch.zeros(B, 3, N), torch.full((B, 3, N), -3.6)
            smp = ps + [torch.from_numpy(src)] if mode == "inverse" else [torch.from_numpy(src)] + ps
            nll = FO.point_flow_nll(smp, [pm] + mus, [pl] + lvs)
            np.testing.assert_allclose(nll.numpy(), gold[c + "/nll"], rtol=2e-5)
    # L=14 truncated stack (the BASELINE metric's layer count)
    st = FO.to_torch(FO.make_decoder_state(7, 5, 64, 128))
    tgt, z, g = FO.synthetic_inputs(7, 2, 128, 128)
    with torch.no_grad():
        ps, mus, lvs = FO.decoder(st, 5, torch.from_numpy(z), torch.from_numpy(g), "direct", n_layers=14)
    assert len(ps) == 14
    np.testing.assert_allclose(ps[-1].numpy(), gold["nf5_L14_direct/final"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(sum(lvs).numpy(), gold["nf5_L14_direct/sum_logvars"], rtol=RTOL, atol=ATOL)


def test_decoder_training_step_vs_reference(golden_dir):
    gold, meta = _load(golden_dir, "flow_decoder")
    case = [c for c in meta["cases"] if c.get("bn") == "train"][0]
    c, n_flows, B, N, G, seed = (case[k] for k in ("tag", "n_flows", "B", "N", "G", "seed"))
    st = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
    for k, v in st.items():
        if v.dtype == torch.float32 and not k.endswith("eps") and "running" not in k:
            v.requires_grad_(True)
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    tp = torch.from_numpy(tgt.copy()).requires_grad_(True)
    tg = torch.from_numpy(g.copy()).requires_grad_(True)
    stats = {}
    ps, mus, lvs = FO.decoder(st, n_flows, tp, tg, "inverse", training=True, stats_out=stats)
    pm, pl = torch.zeros(B, 3, N), torch.full((B, 3, N), -3.6)
    loss = FO.point_flow_nll(ps + [tp], [pm] + mus, [pl] + lvs)
    loss.backward()
    np.testing.assert_allclose(ps[0].detach().numpy(), gold[c + "/ps0"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sum(lvs).detach().numpy(), gold[c + "/sum_logvars"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.detach().numpy(), gold[c + "/nll"], rtol=2e-5)
    np.testing.assert_allclose(tp.grad.numpy(), gold[c + "/grad_p"], rtol=2e-3, atol=2e-3)
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), gold[c + "/stats/" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    named = [(k, v.grad) for k, v in st.items() if v.requires_grad]
    for k, v in _grad_projection(named, seed).items():
        ref = gold[c + "/gproj/" + k]
        assert abs(v[0] - ref[0]) <= 1e-3 * (ref[2] + 1e-6) + 1e-4, (k, v, ref)
        assert abs(v[1] - ref[1]) <= 3e-3 * (ref[2] + 1e-6) + 1e-4, (k, v, ref)


def test_chamfer_oracle_vs_reference_distChamfer(golden_dir):
    gold = np.load(os.path.join(golden_dir, "chamfer.npz"))
    a, b = chamfer_inputs(21, 3, 257, 257)
    d1, i1, d2, i2 = S.nndistance(a, b)
    # reference distChamfer returns (per-b-point, per-a-point) -- the opposite
    # order of NNDistance
    np.testing.assert_allclose(d1, gold["eq257/ref_per_a"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(d2, gold["eq257/ref_per_b"], rtol=1e-4, atol=2e-6)
    for tag, (B, n, m, seed) in {"eq257": (3, 257, 257, 21), "ne": (2, 130, 515, 22)}.items():
        a, b = chamfer_inputs(seed, B, n, m)
        d1, i1, d2, i2 = S.nndistance(a, b)
        np.testing.assert_allclose(d1, gold[tag + "/bf_dist1"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(d2, gold[tag + "/bf_dist2"], rtol=1e-5, atol=1e-7)
        # indices agree with the float64 brute force wherever the best/second-best gap is resolvable in fp32
        ok1 = gold[tag + "/bf_gap1"] > 1e-5
        ok2 = gold[tag + "/bf_gap2"] > 1e-5
        assert ok1.mean() > 0.9 and ok2.mean() > 0.9
        assert np.array_equal(i1[ok1], gold[tag + "/bf_idx1"][ok1])
        assert np.array_equal(i2[ok2], gold[tag + "/bf_idx2"][ok2])


def test_chamfer_oracle_first_minimum_rule():
    # integer-grid coordinates: distances are exact in fp32, ties are exact
    B, n, m = 2, 97, 130
    a = np.round(detrng.uniform_f32(31, (B, n, 3), -3, 3))
    b = np.round(detrng.uniform_f32(32, (B, m, 3), -3, 3))
    b[:, 50:60] = b[:, 10:20]              # duplicate candidates: earlier index must win
    d1, i1, d2, i2 = S.nndistance(a, b)
    dd = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(i1, dd.argmin(2))        # numpy argmin = first minimum
    assert np.array_equal(i2, dd.argmin(1))
    assert np.array_equal(d1, dd.min(2)) and np.array_equal(d2, dd.min(1))
    assert not np.isin(i1, np.arange(50, 60)).any()


def test_chamfer_grad_oracle_matches_autograd():
    B, n, m = 2, 40, 55
    a, b = chamfer_inputs(41, B, n, m)
    d1, i1, d2, i2 = S.nndistance(a, b)
    gd1 = detrng.normal_f32(42, (B, n)); gd2 = detrng.normal_f32(43, (B, m))
    g1, g2 = S.nndistancegrad(a, b, i1, i2, gd1, gd2)
    ta = torch.from_numpy(a).double().requires_grad_(True)
    tb = torch.from_numpy(b).double().requires_grad_(True)
    dd = ((ta[:, :, None, :] - tb[:, None, :, :]) ** 2).sum(-1)
    loss = (dd.min(2)[0] * torch.from_numpy(gd1).double()).sum() + (dd.min(1)[0] * torch.from_numpy(gd2).double()).sum()
    loss.backward()
    np.testing.assert_allclose(g1, ta.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g2, tb.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_emd_oracle_invariants():
    """approx-EMD parity is UNPINNED by the reference (no CPU path / test);
    these are the structural invariants of approxmatch.cu's auction."""
    for (B, n, m) in ((2, 64, 64), (2, 128, 64), (1, 48, 96)):
        a, b = chamfer_inputs(50 + n, B, n, m)
        match, temp = S.approxmatch(a, b)
        multiL, multiR = (1, n // m) if n >= m else (m // n, 1)
        assert match.shape == (B, m, n) and (match >= 0).all()
        assert (match.sum(1) <= multiL * (1 + 1e-4)).all()       # each xyz1 point ships at most multiL
        assert (match.sum(2) <= multiR * (1 + 1e-4)).all()       # each xyz2 point receives at most multiR
        assert match.sum() > 0.95 * min(n * multiL, m * multiR) * B
        cost = S.matchcost(a, b, match)
        assert (cost > 0).all()
        g1, g2 = S.matchcostgrad(a, b, match)
        # d cost / d xyz with match held constant == autograd of sum(match * dist)
        ta = torch.from_numpy(a).double().requires_grad_(True)
        tb = torch.from_numpy(b).double().requires_grad_(True)
        dist = ((ta[:, None, :, :] - tb[:, :, None, :]) ** 2).sum(-1).sqrt()
        (torch.from_numpy(match).double() * dist).sum().backward()
        np.testing.assert_allclose(g1, ta.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(g2, tb.grad.numpy(), rtol=1e-4, atol=1e-5)
    # identical clouds: the matching is (nearly) the identity and the cost ~ 0
    a, _ = chamfer_inputs(60, 2, 64, 64)
    match, _ = S.approxmatch(a, a.copy())
    assert (S.matchcost(a, a.copy(), match) < 0.05).all()
    assert (np.diagonal(match, axis1=1, axis2=2) > 0.9).all()


def test_emd_oracle_vs_an_independent_float64_reading():
    """approx-EMD parity stays UNPINNED by the reference (no CPU path, no vectors: a11-a14 are "partial" for that reason).  What
    CAN be excluded here is a slip in the restatement: a SECOND reading of approxmatch.cu:3-182, written from the source on its
    own -- whole passes as float64 matrix expressions instead of the C oracle's loops -- must give the C oracle's matching.
      level = -4^j, j = 7 .. -1                                                              approxmatch.cu:24-28
      pass 1  ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level d2) remainR[l])               :29-62
      pass 2  sumr = remainR[l] sum_k exp(level d2) ratioL[k];  ratioR[l] = min(remainR[l] / (sumr + 1e-9), 1) remainR[l];
              remainR[l] = max(0, remainR[l] - sumr)                                         :78-111
      pass 3  w = exp(level d2) ratioL[k] ratioR[l];  match[l][k] += w;  remainL[k] = max(0, remainL[k] - sum_l w)   :130-163
    with multiL, multiR = (1, n // m) if n >= m else (m // n, 1)                              :6-12"""
    def reading(a, b):
        n, m = len(a), len(b)
        d2 = ((b[:, None, :].astype(np.float64) - a[None, :, :].astype(np.float64)) ** 2).sum(2)        # (m, n)
        remL = np.full(n, 1.0 if n >= m else float(m // n))
        remR = np.full(m, float(n // m) if n >= m else 1.0)
        match = np.zeros((m, n))
        for j in range(7, -2, -1):
            e = np.exp(-(4.0 ** j) * d2)
            ratioL = remL / (1e-9 + remR @ e)
            sumr = (e @ ratioL) * remR
            ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR
            remR = np.maximum(0.0, remR - sumr)
            w = e * ratioR[:, None] * ratioL[None, :]
            match += w
            remL = np.maximum(0.0, remL - w.sum(0))
        return match
    rng = np.random.default_rng(8)
    for (n, m, kind) in ((64, 64, "uniform"), (128, 64, "uniform"), (48, 96, "uniform"), (200, 77, "line"), (90, 90, "jitter")):
        a = (rng.random((n, 3)) - 0.5).astype(np.float32)
        b = (rng.random((m, 3)) - 0.5).astype(np.float32)
        if kind == "line":
            a[:, 1:] = 0; b[:, 1:] = 0
        if kind == "jitter":
            b = (a[rng.permutation(n)[:m]] + 0.02 * rng.standard_normal((m, 3))).astype(np.float32)
        got, _ = S.approxmatch(a[None], b[None])
        ref = reading(a, b)
        # the C oracle is the CUDA code's fp32; the second reading is float64: the auction amplifies the difference at its clamps
        np.testing.assert_allclose(got[0].sum(0), ref.sum(0), rtol=2e-3, atol=2e-4, err_msg=kind)
        np.testing.assert_allclose(got[0].sum(1), ref.sum(1), rtol=2e-3, atol=2e-4, err_msg=kind)
        assert float(np.abs(got[0] - ref).max()) <= 5e-3, (kind, float(np.abs(got[0] - ref).max()))
        dist = np.sqrt(((b[:, None, :].astype(np.float64) - a[None, :, :]) ** 2).sum(2))
        c_ref = float((ref * dist).sum())
        np.testing.assert_allclose(S.matchcost(a[None], b[None], got)[0], c_ref, rtol=2e-5, err_msg=kind)


def test_encoder_oracle_vs_reference_golden(golden_dir):
    """oracle/encoder_oracle.py against PointNetCloudEncoder + torch.max captured from the reference
    (encoders.py:9-28, models.py:85): eval and train mode outputs, running statistics, d/dx and every parameter
    gradient (projections) through autograd on the restatement."""
    from oracle import encoder_oracle as EO
    gold, meta = _load(golden_dir, "encoder")
    assert meta["keys"] == list(EO.make_encoder_state(0).keys())
    sa, sb = meta["feat_lattice"]
    for case, (seed, B, N) in meta["cases"].items():
        x = torch.from_numpy(EO.encoder_inputs(seed, B, N))
        for training in (False, True):
            tag = "%s_%s" % (case, "train" if training else "eval")
            st = FO.to_torch(EO.make_encoder_state(seed))
            params = {k: v.requires_grad_(True) for k, v in st.items() if v.dtype == torch.float32 and "running" not in k}
            st.update(params)
            xin = x.clone().requires_grad_(training)
            stats = {}
            feat = EO.encoder_features(st, xin, training, stats)
            gmax = torch.max(feat, dim=2)[0]
            np.testing.assert_allclose(gmax.detach().numpy(), gold[tag + "_max"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(feat.detach()[:, ::sa, ::sb].numpy(), gold[tag + "_feat_sub"], rtol=RTOL, atol=ATOL)
            if training:
                r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "enc_r"), tuple(gmax.shape)))
                (gmax * r).sum().backward()
                np.testing.assert_allclose(xin.grad.numpy(), gold[tag + "_dx"], rtol=1e-4, atol=2e-5)   # values O(1)
                proj = _grad_projection([(k, v.grad) for k, v in params.items()], seed)
                for k, v in proj.items():
                    ref = gold[tag + "_gproj_" + k]
                    np.testing.assert_allclose(v, ref, rtol=2e-4, atol=2e-5 * max(1.0, float(ref[2])), err_msg=k)
                for k, v in stats.items():
                    np.testing.assert_allclose(v.numpy(), gold[tag + "_stat_" + k], rtol=RTOL, atol=ATOL, err_msg=k)


def test_optimizer_oracle_and_mirror_vs_reference_golden(golden_dir):
    """oracle/optimizer_oracle.py AND dpf_nets_amd.networks.optimizers (Adam incl. AMSGrad / weight decay, LRUpdater)
    against five steps of the reference's own optimizer (lib/networks/optimizers.py) under its cosine schedule."""
    from oracle import optimizer_oracle as OO
    from oracle.gen_golden import OPT_CASES, optimizer_inputs
    from dpf_nets_amd.networks import optimizers as MO
    gold = np.load(os.path.join(golden_dir, "optimizer.npz"))
    sched_kw = dict(cycle_length=3, min_lr=1e-4, max_lr=2e-3, beta1=0.9, min_beta2=0.99, max_beta2=0.999)
    for name, (ams, wd) in OPT_CASES.items():
        ps = [torch.from_numpy(np.asarray(v)) for v in optimizer_inputs(5, -1)]
        states = [dict() for _ in ps]
        mps = [torch.nn.Parameter(torch.from_numpy(np.asarray(v))) for v in optimizer_inputs(5, -1)]
        opt = MO.Adam(mps, lr=2e-3, weight_decay=wd, betas=(0.9, 0.999), amsgrad=ams)
        sched = MO.LRUpdater(4, **sched_kw)
        for step in range(5):
            lr, betas = OO.lr_update(4, 3, 1e-4, 2e-3, 0.9, 0.99, 0.999, step // 4, step % 4)
            sched(opt, step // 4, step % 4)
            assert opt.param_groups[0]["lr"] == lr and opt.param_groups[0]["betas"] == betas
            np.testing.assert_allclose([lr, betas[1]], gold[name + "_sched"][step], rtol=1e-15)
            gs = [torch.from_numpy(np.asarray(g)) for g in optimizer_inputs(5, step)]
            ps = [OO.adam_step(p, g, st, lr, betas, 1e-8, wd, ams) for p, g, st in zip(ps, gs, states)]
            for p, g in zip(mps, gs):
                p.grad = g
            opt.step()
        for i in range(len(ps)):
            for got, tol in ((ps[i].numpy(), 2e-6), (mps[i].detach().numpy(), 0.0)):    # the mirror runs the same ATen ops
                np.testing.assert_allclose(got, gold["%s_p%d" % (name, i)], rtol=tol, atol=tol * 1e-2)
            st = opt.state[mps[i]]
            assert st["step"] == 5
            np.testing.assert_array_equal(st["exp_avg"].numpy(), gold["%s_m%d" % (name, i)])
            np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), gold["%s_v%d" % (name, i)])
            if ams:
                np.testing.assert_array_equal(st["max_exp_avg_sq"].numpy(), gold["%s_vmax%d" % (name, i)])


def test_gprior_oracle_and_module_vs_reference_golden(golden_dir):
    """oracle/gprior_oracle.py AND the tensor-op path of dpf_nets_amd.networks.GlobalRNVPDecoder against the
    reference's GlobalRNVPDecoder (decoders.py:7-38, flows.py:163-243): the three DIRECT-order lists in both modes,
    eval and train; in train mode d/dg, every parameter gradient (projections) and the BatchNorm running statistics."""
    from oracle import gprior_oracle as GO
    from dpf_nets_amd import networks as nets
    gold, meta = _load(golden_dir, "gprior")
    for case, (seed, n_flows, nf, G, B) in meta["cases"].items():
        g = torch.from_numpy(GO.gprior_inputs(seed, B, G))
        state = GO.make_gprior_state(seed, n_flows, nf, G)
        if case == "a":
            assert meta["keys_case_a"] == list(state.keys())
        for training in (False, True):
            if training and B < 2:
                continue
            for mode in ("direct", "inverse"):
                tag = "%s_%s_%s" % (case, "train" if training else "eval", mode)
                # ---- the oracle
                st = FO.to_torch(state)
                params = {k: v.requires_grad_(True) for k, v in st.items()
                          if v.dtype == torch.float32 and "running" not in k and not k.endswith("eps")}
                st.update(params)
                gin = g.clone().requires_grad_(training)
                stats = {}
                lists = GO.global_rnvp_decoder(st, n_flows, gin, mode, training, stats)
                # ---- the module
                dec = nets.GlobalRNVPDecoder(n_flows, nf, G)
                dec.load_state_dict(FO.to_torch(state), strict=True)
                dec.train(training)
                gmod = g.clone().requires_grad_(training)
                mlists = dec(gmod, mode=mode)
                for name, lst, mlst in zip(("gs", "mus", "lvs"), lists, mlists):
                    assert len(lst) == len(mlst) == 2 * n_flows
                    np.testing.assert_allclose(torch.stack(lst).detach().numpy(), gold[tag + "_" + name], rtol=RTOL, atol=ATOL)
                    np.testing.assert_allclose(torch.stack(list(mlst)).detach().numpy(), gold[tag + "_" + name], rtol=RTOL, atol=ATOL)
                if not training:
                    continue
                for which, ls, gi, named in (("oracle", lists, gin, None), ("module", mlists, gmod, dec)):
                    loss = 0.0
                    for name, lst in zip(("gs", "mus", "lvs"), ls):
                        r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "gprior_r_" + name), (len(lst), B, G)))
                        loss = loss + (torch.stack(list(lst)) * r).sum()
                    loss.backward()
                    np.testing.assert_allclose(gi.grad.numpy(), gold[tag + "_dg"], rtol=2e-4, atol=2e-5, err_msg=which)
                    grads = [(k, v.grad) for k, v in params.items()] if named is None else \
                        [(k, p.grad) for k, p in named.named_parameters()]
                    for k, v in _grad_projection(grads, seed).items():
                        ref = gold[tag + "_gproj_" + k]
                        np.testing.assert_allclose(v, ref, rtol=2e-4, atol=2e-5 * max(1.0, float(ref[2])), err_msg=which + k)
                for k, v in stats.items():
                    np.testing.assert_allclose(v.numpy(), gold[tag + "_stat_" + k], rtol=RTOL, atol=ATOL, err_msg=k)
                for k, v in dec.state_dict().items():
                    if "running" in k:
                        np.testing.assert_allclose(v.numpy(), gold[tag + "_stat_" + k], rtol=RTOL, atol=ATOL, err_msg=k)


def test_model_oracle_evaluating_forward_vs_reference_golden(golden_dir):
    """oracle/model_oracle.evaluating_forward (models.py:173-216 restated) over the CPU oracles reproduces what the
    reference's own Local_Cond_RNVP_MC_Global_RNVP_VAE produced (oracle/check_dropin.py -> tests/golden/model_eval.npz)."""
    from oracle import model_oracle as MO, encoder_oracle as EO, gprior_oracle as GO
    gold = np.load(os.path.join(golden_dir, "model_eval.npz"))
    cfg = MO.CONFIG
    st = FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg))
    x, eps = MO.model_inputs(int(gold["seed"]), int(gold["B"]), int(gold["N"]))
    assert np.array_equal(x, gold["x"]) and np.array_equal(eps, gold["eps"])
    blocks = {
        "pc_encoder": lambda t: EO.encoder_features(FO.sub_state(st, "pc_encoder."), t),
        "g_prior": lambda g, mode: GO.global_rnvp_decoder(FO.sub_state(st, "g_prior."), cfg["g_prior_n_flows"], g, mode),
        "pc_decoder": lambda p, g, mode: FO.decoder(FO.sub_state(st, "pc_decoder."), cfg["p_decoder_n_flows"], p, g, mode),
    }
    with torch.no_grad():
        out = MO.evaluating_forward(blocks, st, torch.from_numpy(x), torch.from_numpy(eps))
    for k in ("g_posterior_mus", "g_posterior_logvars"):
        np.testing.assert_allclose(out[k].numpy(), gold[k], rtol=2e-5, atol=1e-6)
    for k in ("g_prior_samples", "g_prior_logvars", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
        assert len(out[k]) == int(gold[k + "_len"]), k
        for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
            ref = gold["%s/%d" % (k, i)]
            np.testing.assert_allclose(out[k][i].numpy(), ref, rtol=1e-4, atol=2e-6 * max(1.0, float(np.abs(ref).max())), err_msg="%s/%d" % (k, i))
    np.testing.assert_allclose(float(FO.point_flow_nll(out["p_prior_samples"], out["p_prior_mus"], out["p_prior_logvars"])),
                               float(gold["pnll_as_losses_py"]), rtol=1e-5)


def _svr_eval_metrics(final, tgt):
    """evaluating.py:198-205 over the oracle's Chamfer: per-cloud CD and f_score (utils.py:38-42, threshold 0.001)."""
    from oracle import structural as S
    r = np.ascontiguousarray(final.transpose(0, 2, 1))
    t = np.ascontiguousarray(tgt.transpose(0, 2, 1))
    d1, _, d2, _ = S.nndistance(r, t)
    cd = d1.mean(1, dtype=np.float64) + d2.mean(1, dtype=np.float64)
    prec = 100.0 * (d2 < 0.001).mean(1, dtype=np.float64)
    rec = 100.0 * (d1 < 0.001).mean(1, dtype=np.float64)
    return cd, 2.0 * prec * rec / (prec + rec + 1e-7), (d1, d2)


@pytest.mark.parametrize("which", ["small", "predict"])
def test_model_oracle_svr_predicting_forward_vs_reference_golden(golden_dir, which):
    """oracle/model_oracle.predicting_forward (models.py:417-462 restated) over the CPU oracles
    reproduces what the reference's own Local_Cond_RNVP_MC_Global_RNVP_VAE_IC produced in `predicting` mode
    (oracle/check_dropin.py -> tests/golden/model_svr_small.npz; model_svr_predict.npz = the shipped shapes of
    configs/svr/all.yaml: B = 50, 2500 points, G = 512, 63 layers -- there only the first clouds, a few seconds of CPU),
    and the evaluation of evaluating.py:198-205 -- per-cloud CD and f_score -- over the oracle's Chamfer reproduces the
    reference's utils.f_score / distChamfer values."""
    from oracle import model_oracle as MO, gprior_oracle as GO
    gold = np.load(os.path.join(golden_dir, "model_svr_%s.npz" % which))
    cfg = MO.SVR_CONFIG_SMALL if which == "small" else MO.SVR_CONFIG
    seed, B, S = int(gold["seed"]), int(gold["B"]), int(gold["S"])
    st = FO.to_torch(MO.make_svr_state(seed, cfg))
    G = cfg["g_latent_space_size"]
    _, eps, _ = MO.svr_inputs(seed, B, S, G)
    nb = B if which == "small" else gold["final_first"].shape[0]          # the 63-layer CPU oracle on 6 of the 50 clouds
    blocks = {
        "g0_prior": lambda x: MO.feature_encoder(st, "g0_prior", x, cfg["g_prior_n_layers"], False),
        "g_prior": lambda g, mode: GO.global_rnvp_decoder(FO.sub_state(st, "g_prior."), cfg["g_prior_n_flows"], g, mode),
        "p_prior": lambda g: MO.feature_encoder(st, "p_prior", g, cfg["p_prior_n_layers"], True),
        "p_prior_mus": st["p_prior_mus"],
        "pc_decoder": lambda p, g, mode: FO.decoder(FO.sub_state(st, "pc_decoder."), cfg["p_decoder_n_flows"], p, g, mode),
    }
    with torch.no_grad():
        out = MO.predicting_forward(blocks, torch.from_numpy(gold["img_features"][:nb]), torch.from_numpy(eps[:nb]), cfg)
    for k in ("g_prior_samples", "g_prior_mus", "g_prior_logvars"):
        assert len(out[k]) == int(gold[k + "_len"]), k
        for i in (0, 1, len(out[k]) - 1):
            ref = gold["%s/%d" % (k, i)][:nb]
            np.testing.assert_allclose(out[k][i].numpy(), ref, rtol=1e-4, atol=3e-6 * max(1.0, float(np.abs(ref).max())), err_msg="%s/%d" % (k, i))
    assert len(out["p_prior_samples"]) == int(gold["p_prior_samples_len"]) == 3 * cfg["p_decoder_n_flows"] + 1
    np.testing.assert_allclose(out["p_prior_logvars"][0][:, :, 0].numpy(), gold["p_base_logvar"][:nb], rtol=1e-4, atol=1e-5)
    final = out["p_prior_samples"][-1].numpy()
    ref = gold["final_first"][:nb]
    np.testing.assert_allclose(final[:ref.shape[0]], ref, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()))
    np.testing.assert_allclose(sum(out["p_prior_logvars"][1:]).numpy()[:ref.shape[0]], gold["sum_p_logvars_first"][:nb],
                               rtol=1e-4, atol=2e-5 * float(np.abs(gold["sum_p_logvars_first"]).max()))
    np.testing.assert_allclose(np.abs(final).sum((1, 2), dtype=np.float64), gold["final_abs_sum"][:nb], rtol=2e-5)
    tgt = MO.svr_target(seed, B, S, float(gold["target_std"]))[:nb]
    cd, f1, _ = _svr_eval_metrics(final, tgt)
    np.testing.assert_allclose(cd, gold["cd_per_cloud"][:nb], rtol=2e-4)
    # f_score counts points under a threshold: a distance within rounding of 0.001 may fall either side (the reference's
    # distChamfer is the expanded form); one flipped point of S moves precision or recall by 100 / S
    assert np.all(np.abs(f1 - gold["f_score"][:nb]) <= 3 * 100.0 / S + 1e-3 * gold["f_score"][:nb]), (f1, gold["f_score"][:nb])


def test_model_oracle_training_forward_vs_reference_golden(golden_dir):
    """oracle/model_oracle.training_forward + vae_loss (models.py:125-171, losses.py:37-51 restated) over the mirror's
    modules on their CPU tensor-op path reproduce the reference model's training step
    (oracle/check_dropin.py -> tests/golden/model_train.npz): loss terms, outputs, all 250 gradient projections."""
    from oracle import model_oracle as MO
    from oracle.gen_golden import _grad_projection
    from dpf_nets_amd import networks as nets
    gold = np.load(os.path.join(golden_dir, "model_train.npz"))
    cfg = MO.CONFIG
    st = FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg))
    enc = nets.PointNetCloudEncoder(cfg["pc_enc_init_n_channels"], cfg["pc_enc_init_n_features"], cfg["pc_enc_n_features"])
    enc.load_state_dict(FO.sub_state(st, "pc_encoder."), strict=True)
    prior = nets.GlobalRNVPDecoder(cfg["g_prior_n_flows"], cfg["g_prior_n_features"], cfg["g_latent_space_size"])
    prior.load_state_dict(FO.sub_state(st, "g_prior."), strict=True)
    dec = nets.LocalCondRNVPDecoder(cfg["p_decoder_n_flows"], cfg["p_decoder_n_features"], cfg["g_latent_space_size"])
    dec.load_state_dict(FO.sub_state(st, "pc_decoder."), strict=True)
    enc.train(); prior.train(); dec.train()
    gst = {k: v for k, v in st.items() if not k.startswith(("pc_encoder.", "g_prior.", "pc_decoder."))}
    leaf = [k for k in gst if k.startswith(("g_posterior.", "g0_prior"))]
    for k in leaf:
        gst[k].requires_grad_(True)
    x, eps_g = torch.from_numpy(gold["x"]), torch.from_numpy(gold["eps_g"])
    blocks = {"pc_encoder": enc, "g_prior": lambda g, mode: prior(g, mode=mode), "pc_decoder": lambda p, g, mode: dec(p, g, mode=mode)}
    out = MO.training_forward(blocks, gst, x, x, eps_g)
    loss, pnll, gnll, gent = MO.vae_loss(out, nets.PointFlowNLL(), cfg)
    loss.backward()
    got = np.array([float(loss.detach()), float(pnll.detach()), float(gnll.detach()), float(gent.detach())])
    np.testing.assert_allclose(got, gold["loss"], rtol=2e-6)
    for k in ("g_prior_samples", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
        assert len(out[k]) == int(gold[k + "_len"]), k
        for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
            ref = gold["%s/%d" % (k, i)]
            np.testing.assert_allclose(out[k][i].detach().numpy(), ref, rtol=1e-4, atol=2e-6 * max(1.0, float(np.abs(ref).max())),
                                       err_msg="%s/%d" % (k, i))
    grads = {}
    for pre, mod in (("pc_encoder.", enc), ("g_prior.", prior), ("pc_decoder.", dec)):
        for k, p in mod.named_parameters():
            grads[pre + k] = p.grad
    for k in leaf:
        grads[k] = gst[k].grad
    names = [str(k) for k in gold["grad_names"]]
    assert sorted(grads) == sorted(names)
    for k, v in _grad_projection([(k, grads[k]) for k in names], 23).items():
        ref = gold["gradproj/" + k]
        assert np.all(np.abs(v - ref) <= 2e-4 * (ref[2] + 1e-6) + 1e-6), (k, v, ref)


def test_model_mirror_training_step_vs_reference_golden(golden_dir):
    """dpf_nets_amd.networks.Local_Cond_RNVP_MC_Global_RNVP_VAE (+ its loss class) -- the mirror of models.py:13-258 /
    losses.py:37-51 the bench's full-model training leg is built from -- loads the reference model's state dict with
    strict=True and reproduces the reference's training step (tests/golden/model_train.npz) on the CPU tensor-op path."""
    from oracle import model_oracle as MO
    from oracle.gen_golden import _grad_projection
    from dpf_nets_amd import networks as nets
    gold = np.load(os.path.join(golden_dir, "model_train.npz"))
    cfg = dict(MO.CONFIG, util_mode="training")
    model = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg)
    model.load_state_dict(FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg)), strict=True)
    model.train()
    eps_g = torch.from_numpy(gold["eps_g"])
    model.reparameterize = lambda mu, logvar: eps_g * torch.exp(0.5 * logvar) + mu
    x = torch.from_numpy(gold["x"])
    out = model(x, x)
    loss, pnll, gnll, gent = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)(x, x, out)
    loss.backward()
    np.testing.assert_allclose(np.array([float(v.detach()) for v in (loss, pnll, gnll, gent)]), gold["loss"], rtol=2e-6)
    names = [str(k) for k in gold["grad_names"]]
    assert [k for k, _ in model.named_parameters()] == names
    for k, v in _grad_projection([(k, p.grad) for k, p in model.named_parameters()], 23).items():
        ref = gold["gradproj/" + k]
        assert np.all(np.abs(v - ref) <= 2e-4 * (ref[2] + 1e-6) + 1e-6), (k, v, ref)
    for k, b in model.named_buffers():
        np.testing.assert_allclose(b.detach().numpy(), gold["buffer/" + k], rtol=1e-5, atol=1e-7, err_msg=k)
    # the other configurations construct with the reference's parameter counts (12 972 413 for all_scaled)
    big = dict(cfg, g_latent_space_size=512, g_prior_n_flows=7, g_prior_n_features=128, g_posterior_n_layers=1, p_prior_n_layers=1,
               p_decoder_n_flows=21, p_decoder_base_type="freevar")
    assert sum(p.numel() for p in nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**big).parameters()) == 12972413
