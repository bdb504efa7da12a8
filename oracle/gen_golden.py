"""Generate tests/golden/*.npz by running the REFERENCE's own PyTorch modules on CPU.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden

The reference never travels: only the vectors written here do.  Inputs and
weights are re-derivable from seeds (oracle/detrng.py), so the fixtures hold
outputs only.  TEST INFRASTRUCTURE.
"""
import json
import os
import sys
import types

import numpy as np
import torch

from . import detrng
from . import flow_oracle as FO
from . import encoder_oracle as EO
from . import gprior_oracle as GO

REF = os.environ.get("DPF_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

WARPS = ([0], [1], [2], [0, 1], [0, 2], [1, 2])


def _import_reference():
    sys.path.insert(0, REF)
    # lib.metrics.evaluation_metrics imports the CUDA extension at module scope
    # (evaluation_metrics.py:9-10); stub it so the pure-PyTorch distChamfer
    # (:35-45) becomes importable.  Nothing from the stub is ever called.
    for name in ("lib.metrics.StructuralLosses", "lib.metrics.StructuralLosses.match_cost",
                 "lib.metrics.StructuralLosses.nn_distance"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["lib.metrics.StructuralLosses.match_cost"].match_cost = None
    sys.modules["lib.metrics.StructuralLosses.nn_distance"].nn_distance = None
    from lib.networks import flows, decoders, losses, layers
    from lib.metrics import evaluation_metrics
    return flows, decoders, losses, layers, evaluation_metrics


def _grad_projection(named_grads, seed):
    """Two scalars per parameter gradient: its sum and its dot with a fixed
    pseudo-random vector -- small, and sensitive to any permutation/scale bug."""
    out = {}
    for k, gr in named_grads:
        v = gr.detach().numpy().astype(np.float64).ravel()
        r = detrng.normal(detrng.key(seed, "proj:" + k), v.size)
        out[k] = np.array([v.sum(), float(v @ r), float(np.abs(v).sum())])
    return out


def layer_inputs(seed, B, N, G):
    p = detrng.normal_f32(detrng.key(seed, "p"), (B, 3, N), 0.0, 0.3)
    g = detrng.normal_f32(detrng.key(seed, "g"), (B, G), 0.0, 1.0)
    r1 = detrng.normal_f32(detrng.key(seed, "r1"), (B, 3, N))
    r2 = detrng.normal_f32(detrng.key(seed, "r2"), (B, 3, N))
    r3 = detrng.normal_f32(detrng.key(seed, "r3"), (B, 3, N))
    return p, g, r1, r2, r3


def gen_layer(flows):
    B, N, F, G = 4, 256, 64, 128
    out = {}
    meta = {"B": B, "N": N, "F": F, "G": G, "cases": []}
    for wi, warp in enumerate(WARPS):
        seed = 100 + wi
        state = FO.make_layer_state(seed, F, G, warp)
        p, g, r1, r2, r3 = layer_inputs(seed, B, N, G)
        for mode in ("direct", "inverse"):
            for bn in ("eval", "train"):
                mod = flows.CondRealNVPFlow3D(F, G, weight_std=0.01, warp_inds=list(warp))
                mod.load_state_dict(FO.to_torch(state), strict=True)
                mod.train(bn == "train")
                tp = torch.from_numpy(p.copy()).requires_grad_(True)
                tg = torch.from_numpy(g.copy()).requires_grad_(True)
                p_out, mu, lv = mod(tp, tg, mode=mode)
                tag = "w%s_%s_%s" % ("".join(map(str, warp)), mode, bn)
                out[tag + "/p_out"] = p_out.detach().numpy()
                out[tag + "/mu"] = mu.detach().numpy()
                out[tag + "/logvar"] = lv.detach().numpy()
                # gradients of a fixed linear functional of all three outputs
                loss = (p_out * torch.from_numpy(r1)).sum() + (lv * torch.from_numpy(r2)).sum() \
                    + (mu * torch.from_numpy(r3)).sum()
                loss.backward()
                out[tag + "/grad_p"] = tp.grad.numpy()
                out[tag + "/grad_g"] = tg.grad.numpy()
                proj = _grad_projection([(k, v.grad) for k, v in mod.named_parameters()], seed)
                for k, v in proj.items():
                    out[tag + "/gproj/" + k] = v
                if bn == "train":
                    sd = mod.state_dict()
                    for k, v in sd.items():
                        if k.endswith("running_mean") or k.endswith("running_var"):
                            out[tag + "/stats/" + k] = v.numpy()
                meta["cases"].append({"tag": tag, "warp": list(warp), "mode": mode, "bn": bn, "seed": seed})
    np.savez_compressed(os.path.join(OUT, "flow_layer.npz"), **out)
    with open(os.path.join(OUT, "flow_layer.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_decoder(decoders, losses):
    out = {}
    meta = {"cases": []}
    nll = losses.PointFlowNLL()
    for (tag, n_flows, B, N, G, seed) in (("nf5", 5, 2, 128, 128, 7), ("nf21", 21, 2, 64, 128, 8),
                                           ("nf2_g512", 2, 3, 100, 512, 9)):
        F = 64
        state = FO.make_decoder_state(seed, n_flows, F, G)
        dec = decoders.LocalCondRNVPDecoder(n_flows, F, G, weight_std=0.01)
        dec.load_state_dict(FO.to_torch(state), strict=True)
        dec.eval()
        tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
        L = 3 * n_flows
        picks = sorted(set([0, min(7, L - 1), L - 1]))
        with torch.no_grad():
            for mode, src in (("direct", z), ("inverse", tgt)):
                ps, mus, lvs = dec(torch.from_numpy(src), torch.from_numpy(g), mode=mode)
                assert len(ps) == len(mus) == len(lvs) == L
                c = "%s_%s" % (tag, mode)
                for k in picks:
                    out["%s/ps%d" % (c, k)] = ps[k].numpy()
                    out["%s/mus%d" % (c, k)] = mus[k].numpy()
                    out["%s/logvars%d" % (c, k)] = lvs[k].numpy()
                out[c + "/sum_logvars"] = sum(lvs).numpy()
                # the loss as models.py:152-171 assembles it: [prior] + decoder lists
                prior_mu = torch.zeros(B, 3, N)
                prior_lv = torch.full((B, 3, N), -3.6)
                smp = ps + [torch.from_numpy(src)] if mode == "inverse" else [torch.from_numpy(src)] + ps
                out[c + "/nll"] = nll(smp, [prior_mu] + mus, [prior_lv] + lvs).numpy()
                meta["cases"].append({"tag": c, "n_flows": n_flows, "B": B, "N": N, "G": G, "seed": seed,
                                      "mode": mode, "picks": picks})
        if tag == "nf5":
            # the BASELINE metric's L=14: first 14 direct-order layers of n_flows=5
            with torch.no_grad():
                cur = torch.from_numpy(z)
                tg = torch.from_numpy(g)
                lv_sum = torch.zeros_like(cur)
                layers = []
                for tri in dec.flows:
                    layers += [tri.nvp1, tri.nvp2, tri.nvp3]
                for lyr in layers[:14]:
                    cur, _, lv = lyr(cur, tg, mode="direct")
                    lv_sum = lv_sum + lv
                out["nf5_L14_direct/final"] = cur.numpy()
                out["nf5_L14_direct/sum_logvars"] = lv_sum.numpy()
    # training-mode decoder: inverse + NLL + backward (training.py:37-55 path)
    n_flows, B, N, G, F, seed = 2, 4, 96, 128, 64, 11
    state = FO.make_decoder_state(seed, n_flows, F, G)
    dec = decoders.LocalCondRNVPDecoder(n_flows, F, G, weight_std=0.01)
    dec.load_state_dict(FO.to_torch(state), strict=True)
    dec.train()
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    tp = torch.from_numpy(tgt.copy()).requires_grad_(True)
    tg = torch.from_numpy(g.copy()).requires_grad_(True)
    ps, mus, lvs = dec(tp, tg, mode="inverse")
    prior_mu = torch.zeros(B, 3, N)
    prior_lv = torch.full((B, 3, N), -3.6)
    loss = nll(ps + [tp], [prior_mu] + mus, [prior_lv] + lvs)
    loss.backward()
    c = "train_nf2_inverse"
    out[c + "/ps0"] = ps[0].detach().numpy()
    out[c + "/sum_logvars"] = sum(lvs).detach().numpy()
    out[c + "/nll"] = loss.detach().numpy()
    out[c + "/grad_p"] = tp.grad.numpy()
    out[c + "/grad_g"] = tg.grad.numpy()
    for k, v in _grad_projection([(k, v.grad) for k, v in dec.named_parameters()], seed).items():
        out[c + "/gproj/" + k] = v
    for k, v in dec.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out[c + "/stats/" + k] = v.numpy()
    meta["cases"].append({"tag": c, "n_flows": n_flows, "B": B, "N": N, "G": G, "seed": seed, "mode": "inverse",
                          "bn": "train"})
    np.savez_compressed(os.path.join(OUT, "flow_decoder.npz"), **out)
    with open(os.path.join(OUT, "flow_decoder.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_state_keys(flows, decoders, layers):
    keys = {}
    for warp in ([0], [0, 1]):
        mod = flows.CondRealNVPFlow3D(64, 128, warp_inds=list(warp))
        keys["CondRealNVPFlow3D_w" + "".join(map(str, warp))] = [
            [k, list(v.shape), str(v.dtype)] for k, v in mod.state_dict().items()]
        keys["CondRealNVPFlow3D_w" + "".join(map(str, warp)) + "_params"] = [k for k, _ in mod.named_parameters()]
    dec = decoders.LocalCondRNVPDecoder(2, 64, 512)
    keys["LocalCondRNVPDecoder_nf2_g512"] = [[k, list(v.shape), str(v.dtype)] for k, v in dec.state_dict().items()]
    keys["LocalCondRNVPDecoder_nf2_g512_nparams"] = sum(p.numel() for p in dec.parameters())
    # init statistics the mirror must reproduce (flows.py:52-58, layers.py:29-38)
    torch.manual_seed(0)
    mod = flows.CondRealNVPFlow3D(64, 128, weight_std=0.01, warp_inds=[0])
    keys["init_stats"] = {
        "sd0_absmax": float(mod.T_mu_0[0].weight.abs().max()),
        "sd1_absmax": float(mod.T_mu_0[3].weight.abs().max()),
        "sd2_std": float(mod.T_mu_1[-1].weight.std()),
        "sd2_bias_absmax": float(mod.T_mu_1[-1].bias.abs().max()),
        "film1_std": float(mod.T_mu_0_cond_w[-1].weight.std()),
        "film1_bias_absmax": float(mod.T_mu_0_cond_w[-1].bias.abs().max()),
    }
    sd = layers.SharedDot(3, 5, 1, bias=True)
    keys["SharedDot_3_5_bias"] = [[k, list(v.shape)] for k, v in sd.state_dict().items()]
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        json.dump(keys, f, indent=1)


def chamfer_inputs(seed, B, n, m):
    a = detrng.uniform_f32(detrng.key(seed, "a"), (B, n, 3), -0.5, 0.5)
    b = detrng.uniform_f32(detrng.key(seed, "b"), (B, m, 3), -0.5, 0.5)
    return a, b


def gen_chamfer(evaluation_metrics):
    out = {}
    # (1) the reference's pure-PyTorch distChamfer (n == m only, :41-43)
    a, b = chamfer_inputs(21, 3, 257, 257)
    with torch.no_grad():
        r0, r1 = evaluation_metrics.distChamfer(torch.from_numpy(a), torch.from_numpy(b))
    # distChamfer returns (P.min(1), P.min(2)) = (per-b-point, per-a-point): evaluation_metrics.py:44
    out["eq257/ref_per_b"] = r0.numpy()
    out["eq257/ref_per_a"] = r1.numpy()
    # (2) float64 brute force for n != m (and for the case above)
    for tag, (B, n, m, seed) in {"eq257": (3, 257, 257, 21), "ne": (2, 130, 515, 22)}.items():
        a, b = chamfer_inputs(seed, B, n, m)
        d = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :].astype(np.float64)) ** 2).sum(-1)
        out[tag + "/bf_dist1"] = d.min(2)
        out[tag + "/bf_idx1"] = d.argmin(2).astype(np.int32)
        out[tag + "/bf_dist2"] = d.min(1)
        out[tag + "/bf_idx2"] = d.argmin(1).astype(np.int32)
        srt = np.sort(d, axis=2)
        out[tag + "/bf_gap1"] = srt[:, :, 1] - srt[:, :, 0]
        srt = np.sort(d, axis=1)
        out[tag + "/bf_gap2"] = srt[:, 1, :] - srt[:, 0, :]
    np.savez_compressed(os.path.join(OUT, "chamfer.npz"), **out)


def gen_encoder():
    """PointNetCloudEncoder (encoders.py:9-28) + the models' max over the points (models.py:85), eval and train mode.
    Shapes: a ragged N (not a multiple of 32 or 256) and a multi-workgroup one."""
    from lib.networks import encoders
    out = {}
    for case, (seed, B, N) in {"a": (11, 2, 300), "b": (12, 3, 1000)}.items():
        st = EO.make_encoder_state(seed)
        x = torch.from_numpy(EO.encoder_inputs(seed, B, N))
        for training in (False, True):
            enc = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
            enc.load_state_dict(FO.to_torch(st), strict=True)
            enc.train(training)
            xin = x.clone().requires_grad_(training)
            feat = enc(xin)
            gmax, gidx = torch.max(feat, dim=2)
            tag = "%s_%s" % (case, "train" if training else "eval")
            out[tag + "_max"] = gmax.detach().numpy()
            out[tag + "_feat_sub"] = feat.detach()[:, ::37, ::29].contiguous().numpy()     # a lattice of the per-point features
            if training:
                r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "enc_r"), tuple(gmax.shape)))
                (gmax * r).sum().backward()
                out[tag + "_dx"] = xin.grad.numpy()
                for k, v in _grad_projection([(k, p.grad) for k, p in enc.named_parameters()], seed).items():
                    out[tag + "_gproj_" + k] = v
                for k, v in enc.state_dict().items():
                    if "running" in k:
                        out[tag + "_stat_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)
    with open(os.path.join(OUT, "encoder.json"), "w") as f:
        json.dump({"cases": {"a": [11, 2, 300], "b": [12, 3, 1000]}, "feat_lattice": [37, 29],
                   "keys": list(enc.state_dict().keys())}, f, indent=1)


OPT_SHAPES = ((1, 64, 3), (64,), (7, 5), ())
OPT_CASES = {"plain": (False, 0.0), "ams_wd": (True, 1e-6), "ams": (True, 0.0), "wd": (False, 1e-2)}


def optimizer_inputs(seed, step):
    """parameters (step = -1) or the gradients of a step, seeded"""
    tag = "p" if step < 0 else "g%d" % step
    return [detrng.normal_f32(detrng.key(seed, "%s:%d" % (tag, i)), shp, 0.0, 1.0 if step < 0 else 0.3) for i, shp in enumerate(OPT_SHAPES)]


def gen_optimizer():
    """lib/networks/optimizers.py: five Adam steps on four tensors under the cosine LRUpdater schedule, for the four
    (amsgrad, weight_decay) variants; final parameters and moment buffers."""
    from lib.networks import optimizers
    out = {}
    sched_kw = dict(cycle_length=3, min_lr=1e-4, max_lr=2e-3, beta1=0.9, min_beta2=0.99, max_beta2=0.999)
    for name, (ams, wd) in OPT_CASES.items():
        params = [torch.nn.Parameter(torch.from_numpy(np.asarray(v))) for v in optimizer_inputs(5, -1)]
        opt = optimizers.Adam(params, lr=2e-3, weight_decay=wd, betas=(0.9, 0.999), amsgrad=ams)
        sched = optimizers.LRUpdater(4, **sched_kw)
        lrs = []
        for step in range(5):
            sched(opt, step // 4, step % 4)
            lrs.append([opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][1]])
            for p, g in zip(params, optimizer_inputs(5, step)):
                p.grad = torch.from_numpy(np.asarray(g))
            opt.step()
        for i, p in enumerate(params):
            out["%s_p%d" % (name, i)] = p.detach().numpy()
            out["%s_m%d" % (name, i)] = opt.state[p]["exp_avg"].numpy()
            out["%s_v%d" % (name, i)] = opt.state[p]["exp_avg_sq"].numpy()
            if ams:
                out["%s_vmax%d" % (name, i)] = opt.state[p]["max_exp_avg_sq"].numpy()
        out[name + "_sched"] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "optimizer.npz"), **out)


GPRIOR_CASES = {"a": (21, 2, 32, 16, 5), "b": (22, 7, 128, 128, 4), "c": (23, 1, 128, 512, 3), "d": (24, 3, 48, 10, 1)}


def gen_gprior(decoders):
    """GlobalRNVPDecoder (decoders.py:7-38) on (B,G) latents, both modes: the three DIRECT-order lists in eval mode;
    in train mode also d/dg of a seeded projection of the lists, projections of the parameter gradients and the
    BatchNorm running statistics.  Cases: a toy, the generation configs' 7 x 128 on G=128, G=512, and a single row."""
    out = {}
    for case, (seed, n_flows, nf, G, B) in GPRIOR_CASES.items():
        st = GO.make_gprior_state(seed, n_flows, nf, G)
        g = torch.from_numpy(GO.gprior_inputs(seed, B, G))
        for training in (False, True):
            if training and B < 2:
                continue                                      # BatchNorm1d refuses a single row in train mode
            for mode in ("direct", "inverse"):
                dec = decoders.GlobalRNVPDecoder(n_flows, nf, G)
                dec.load_state_dict(FO.to_torch(st), strict=True)
                dec.train(training)
                gin = g.clone().requires_grad_(training)
                gs, mus, lvs = dec(gin, mode=mode)
                tag = "%s_%s_%s" % (case, "train" if training else "eval", mode)
                out[tag + "_gs"] = torch.stack(gs).detach().numpy()
                out[tag + "_mus"] = torch.stack(mus).detach().numpy()
                out[tag + "_lvs"] = torch.stack(lvs).detach().numpy()
                if training:
                    loss = 0.0
                    for name, lst in (("gs", gs), ("mus", mus), ("lvs", lvs)):
                        r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "gprior_r_" + name), (len(lst), B, G)))
                        loss = loss + (torch.stack(lst) * r).sum()
                    loss.backward()
                    out[tag + "_dg"] = gin.grad.numpy()
                    for k, v in _grad_projection([(k, p.grad) for k, p in dec.named_parameters()], seed).items():
                        out[tag + "_gproj_" + k] = v
                    for k, v in dec.state_dict().items():
                        if "running" in k:
                            out[tag + "_stat_" + k] = v.numpy()
        if case == "a":
            keys = list(dec.state_dict().keys())
    np.savez_compressed(os.path.join(OUT, "gprior.npz"), **out)
    with open(os.path.join(OUT, "gprior.json"), "w") as f:
        json.dump({"cases": {k: list(v) for k, v in GPRIOR_CASES.items()}, "keys_case_a": keys}, f, indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    flows, decoders, losses, layers, evaluation_metrics = _import_reference()
    only = set(sys.argv[1:])                                  # e.g. `python -m oracle.gen_golden gprior`
    todo = (("keys", lambda: gen_state_keys(flows, decoders, layers)), ("layer", lambda: gen_layer(flows)),
            ("decoder", lambda: gen_decoder(decoders, losses)), ("chamfer", lambda: gen_chamfer(evaluation_metrics)),
            ("encoder", gen_encoder), ("optimizer", gen_optimizer), ("gprior", lambda: gen_gprior(decoders)))
    for name, fn in todo:
        if not only or name in only:
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
