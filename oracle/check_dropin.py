#!/usr/bin/env python3
"""Drop-in proof against the reference's own CALLERS (build container only: imports /root/reference).

1. Imports the reference's lib/networks/models.py twice: as it is, and with the classes INTEGRATION.md section 2 tells a
   maintainer to swap (SharedDot/Swish, CondRealNVPFlow3D(Triple), RealNVPFlow(Couple), LocalCondRNVPDecoder,
   GlobalRNVPDecoder, PointNetCloudEncoder, PointFlowNLL) taken from dpf_nets_amd.networks.  Both models load the same state
   dict (strict), run `Local_Cond_RNVP_MC_Global_RNVP_VAE.forward` in TRAINING mode on CPU (the mirror's tensor-op path)
   under the same torch generator seed, and every entry of the output dict, the loss terms
   (Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss, losses.py:37-51) and every parameter gradient are compared.
2. Runs the reference model in EVALUATING mode (models.py:173-216) with reparameterize's noise replaced by a fixed eps and
   writes tests/golden/model_eval.npz -- what tests/test_gpu_model.py reproduces through the HIP path with the mirror
   classes only, and tests/test_oracle_golden.py through the CPU oracles.
3. The same for one TRAINING step (tests/golden/model_train.npz).
4. (r04) The single-view-reconstruction model `Local_Cond_RNVP_MC_Global_RNVP_VAE_IC` (models.py:261-464): training mode,
   reference vs the model on the swapped classes (CPU); `predicting` mode + the evaluation of evaluating.py:198-205 (CD,
   f_score) by the reference model -> tests/golden/model_svr_small.npz and, at the shipped shapes of configs/svr/all.yaml
   (B = 50, 2500 points, G = 512, 63 layers), tests/golden/model_svr_predict.npz.  The ResNet stays the reference's; the
   fixtures carry its output.
Usage: python oracle/check_dropin.py [--write]
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DPF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import flow_oracle as FO          # noqa: E402
from oracle import model_oracle as MO         # noqa: E402


def _load(pkg, name, path):
    spec = importlib.util.spec_from_file_location(pkg + "." + name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[pkg + "." + name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(pkg, swap):
    """The reference's lib/networks/{models,losses}.py as package `pkg`; with `swap` the building blocks come from the mirror."""
    net = os.path.join(REF, "lib", "networks")
    for p in (pkg.split(".")[0], pkg):
        m = types.ModuleType(p)
        m.__path__ = []
        sys.modules[p] = m
    layers = _load(pkg, "layers", os.path.join(net, "layers.py"))
    flows = _load(pkg, "flows", os.path.join(net, "flows.py"))
    decoders = _load(pkg, "decoders", os.path.join(net, "decoders.py"))
    encoders = _load(pkg, "encoders", os.path.join(net, "encoders.py"))
    _load(pkg, "resnet", os.path.join(net, "resnet.py"))
    losses = _load(pkg, "losses", os.path.join(net, "losses.py"))
    if swap:                                     # INTEGRATION.md section 2, line for line
        from dpf_nets_amd import networks as M
        from dpf_nets_amd.networks import layers as ML, flows as MF, prior_flows as MP, decoders as MD, encoders as ME, losses as MLo
        layers.SharedDot, layers.Swish = ML.SharedDot, ML.Swish
        encoders.SharedDot, encoders.Swish = ML.SharedDot, ML.Swish      # encoders.py:6 `from .layers import SharedDot, Swish`
        flows.CondRealNVPFlow3D, flows.CondRealNVPFlow3DTriple = MF.CondRealNVPFlow3D, MF.CondRealNVPFlow3DTriple
        flows.RealNVPFlow, flows.RealNVPFlowCouple = MP.RealNVPFlow, MP.RealNVPFlowCouple
        decoders.LocalCondRNVPDecoder, decoders.GlobalRNVPDecoder = MD.LocalCondRNVPDecoder, MP.GlobalRNVPDecoder
        encoders.PointNetCloudEncoder = ME.PointNetCloudEncoder
        losses.PointFlowNLL = MLo.PointFlowNLL
        assert M.LocalCondRNVPDecoder is MD.LocalCondRNVPDecoder
    models = _load(pkg, "models", os.path.join(net, "models.py"))     # binds the (possibly swapped) names at import
    return models, losses


def build(models, losses, cfg, state):
    model = models.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg)
    model.load_state_dict(FO.to_torch(state), strict=True)
    return model, losses.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)


def flat(v):
    if torch.is_tensor(v):
        return [v]
    return [t for x in v for t in flat(x)]


def main():
    write = "--write" in sys.argv
    cfg = dict(MO.CONFIG)
    state = MO.make_model_state(11, cfg)
    B, N = 4, 96
    x, eps = MO.model_inputs(11, B, N)
    tx = torch.from_numpy(x)
    torch.set_num_threads(4)

    ref_models, ref_losses = load_reference("libref.networks", swap=False)
    mir_models, mir_losses = load_reference("libmir.networks", swap=True)

    # ---- 1. training mode: reference model vs the same model on the mirror's classes ------------------------
    worst = 0.0
    results = []
    for tag, (models, losses) in (("reference", (ref_models, ref_losses)), ("mirror", (mir_models, mir_losses))):
        cfg_t = dict(cfg, util_mode="training")
        model, loss_fn = build(models, losses, cfg_t, state)
        model.train()
        torch.manual_seed(5)
        out = model(tx, tx)
        loss, pnll, gnll, gent = loss_fn(tx, tx, out)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        bufs = {k: b.clone() for k, b in model.named_buffers()}
        results.append((out, (loss, pnll, gnll, gent), grads, bufs, type(model.pc_decoder).__module__))
    (o1, l1, g1, b1, m1), (o2, l2, g2, b2, m2) = results
    assert m1.startswith("libref") and m2.startswith("dpf_nets_amd"), (m1, m2)       # the swap really took place
    assert set(o1) == set(o2), set(o1) ^ set(o2)
    for k in sorted(o1):
        a, b = flat(o1[k]), flat(o2[k])
        assert len(a) == len(b), (k, len(a), len(b))
        for i, (u, v) in enumerate(zip(a, b)):
            assert u.shape == v.shape, (k, i, u.shape, v.shape)
            err = float((u.detach() - v.detach()).abs().max() / (u.detach().abs().max() + 1e-30))
            worst = max(worst, err)
            assert err < 2e-5, ("output", k, i, err)
    for u, v, name in zip(l1, l2, ("loss", "pnll", "gnll", "gent")):
        assert abs(float(u) - float(v)) <= 2e-5 * max(1.0, abs(float(u))), (name, float(u), float(v))
    assert set(g1) == set(g2)
    for k in g1:
        err = float((g1[k] - g2[k]).abs().max() / (g1[k].abs().max() + 1e-30))
        assert err < 2e-3, ("grad", k, err)
    for k in b1:                                   # BatchNorm running statistics after the step
        assert torch.allclose(b1[k].float(), b2[k].float(), rtol=1e-4, atol=1e-6), ("buffer", k)
    print("training mode: %d output entries, 4 loss terms, %d gradients, %d buffers agree (worst output rel err %.2e)"
          % (len(o1), len(g1), len(b1), worst))

    # ---- 2. evaluating mode golden from the REFERENCE model (fixed eps instead of torch.randn_like) ----------
    model, _ = build(ref_models, ref_losses, cfg, state)
    model.eval()
    teps = torch.from_numpy(eps)
    model.reparameterize = lambda mu, logvar: teps * torch.exp(0.5 * logvar) + mu        # models.py:76-79 with a fixed eps
    with torch.no_grad():
        out = model(tx, tx)
    gold = {"x": x, "eps": eps, "seed": np.array(11), "B": np.array(B), "N": np.array(N)}
    for k in ("g_posterior_mus", "g_posterior_logvars"):
        gold[k] = out[k].detach().numpy()
    for k in ("g_prior_samples", "g_prior_logvars", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
        gold[k + "_len"] = np.array(len(out[k]))
        for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
            gold["%s/%d" % (k, i)] = np.ascontiguousarray(out[k][i].detach().numpy())
    gold["sum_p_logvars"] = sum(out["p_prior_logvars"]).detach().numpy()
    pnll = ref_losses.PointFlowNLL()(out["p_prior_samples"], out["p_prior_mus"], out["p_prior_logvars"])
    gold["pnll_as_losses_py"] = np.array(float(pnll))
    # reconstruction Chamfer as evaluating.py:110-113 reduces it, with the reference's pure-PyTorch distChamfer
    sys.modules.setdefault("lib", types.ModuleType("lib"))
    for name in ("lib.metrics", "lib.metrics.StructuralLosses", "lib.metrics.StructuralLosses.match_cost",
                 "lib.metrics.StructuralLosses.nn_distance"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["lib.metrics.StructuralLosses.match_cost"].match_cost = None
    sys.modules["lib.metrics.StructuralLosses.nn_distance"].nn_distance = None
    em = _load("lib.metrics", "evaluation_metrics", os.path.join(REF, "lib", "metrics", "evaluation_metrics.py"))
    r = out["p_prior_samples"][-1].transpose(1, 2).contiguous()
    t = tx.transpose(1, 2).contiguous()
    dl, dr = em.distChamfer(r, t)
    gold["cd_per_cloud"] = (dl.mean(1) + dr.mean(1)).numpy()
    write_or_check(gold, os.path.join(ROOT, "tests", "golden", "model_eval.npz"), write, "evaluating mode")

    # ---- 3. TRAINING-mode golden from the REFERENCE model: one step of training.py:37-55 (forward, the four loss terms of
    # losses.py:37-51, backward) with reparameterize's noise replaced by a fixed eps_g -- what tests/test_gpu_model.py
    # reproduces through the HIP training kernels (encoder, prior flow, decoder, NLL) with the mirror classes only.
    # Big enough for the HIP path's tiles (B*N a few thousand points), small enough for seconds of CPU.
    Bt, Nt = 6, 640
    xt, _ = MO.model_inputs(23, Bt, Nt)
    from oracle import detrng
    eps_g = detrng.normal_f32(detrng.key(23, "eps_g"), (Bt, cfg["g_latent_space_size"]), 0.0, 1.0)
    cfg_t = dict(cfg, util_mode="training")
    model, loss_fn = build(ref_models, ref_losses, cfg_t, state)
    model.train()
    teg = torch.from_numpy(eps_g)
    model.reparameterize = lambda mu, logvar: teg * torch.exp(0.5 * logvar) + mu
    txt = torch.from_numpy(xt)
    out = model(txt, txt)
    loss, pnll, gnll, gent = loss_fn(txt, txt, out)
    loss.backward()
    tg = {"x": xt, "eps_g": eps_g, "seed": np.array(11), "B": np.array(Bt), "N": np.array(Nt),
          "loss": np.array([float(loss), float(pnll), float(gnll), float(gent)], np.float64)}
    for k in ("g_posterior_mus", "g_posterior_logvars", "g_posterior_samples"):
        tg[k] = out[k].detach().numpy()
    for k in ("g_prior_samples", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
        tg[k + "_len"] = np.array(len(out[k]))
        for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
            tg["%s/%d" % (k, i)] = np.ascontiguousarray(out[k][i].detach().numpy())
    tg["sum_p_logvars"] = sum(out["p_prior_logvars"]).detach().numpy()
    from oracle.gen_golden import _grad_projection
    named = [(k, p.grad) for k, p in model.named_parameters()]
    assert all(g is not None for _, g in named)
    for k, v in _grad_projection(named, 23).items():
        tg["gradproj/" + k] = v
    tg["grad_names"] = np.array([k for k, _ in named])
    for k, b in model.named_buffers():                   # BatchNorm running statistics / counters after the step
        tg["buffer/" + k] = b.detach().numpy().copy()
    print("training golden: loss %.6f pnll %.6f gnll %.6f gent %.6f, %d gradients, %d buffers"
          % (float(loss), float(pnll), float(gnll), float(gent), len(named), sum(1 for _ in model.named_buffers())))
    write_or_check(tg, os.path.join(ROOT, "tests", "golden", "model_train.npz"), write, "training mode")
    svr_sections(ref_models, ref_losses, mir_models, mir_losses, em, write)


def _reference_utils(em):
    """The reference's lib/networks/utils.py (f_score, distChamferCUDA) with its CUDA-only nn_distance served by the
    reference's own pure-PyTorch distChamfer (evaluation_metrics.py:35-45; it returns (per-y, per-x): swapped here into
    nn_distance's (dist1 per x, dist2 per y) order)."""
    def nn_distance(x, y):
        per_y, per_x = em.distChamfer(x, y)
        return per_x, per_y
    sys.modules["lib.metrics.StructuralLosses.nn_distance"].nn_distance = nn_distance
    sys.modules.setdefault("lib.networks", types.ModuleType("lib.networks"))
    return _load("lib.networks", "utils", os.path.join(REF, "lib", "networks", "utils.py"))


def _build_ic(models, cfg, state, resnet_state=None):
    torch.manual_seed(3)                                   # the ResNet keeps its own (seeded) default initialisation
    model = models.Local_Cond_RNVP_MC_Global_RNVP_VAE_IC(**cfg)
    res = model.load_state_dict(FO.to_torch(state), strict=False)
    assert not res.unexpected_keys and all(k.startswith("img_encoder.") for k in res.missing_keys), res
    if resnet_state is not None:
        model.img_encoder.load_state_dict(resnet_state, strict=True)
    return model


def _predict_golden(model, ref_utils, cfg, seed, B, S, img_hw, keep_clouds):
    """The reference IC model in `predicting` mode (models.py:417-462) + the evaluation of evaluating.py:198-205."""
    from oracle import detrng
    G = cfg["g_latent_space_size"]
    _, eps, _ = MO.svr_inputs(seed, B, S, G)
    images = detrng.normal_f32(detrng.key(seed, "images"), (B, 4, img_hw, img_hw), 0.0, 1.0)
    teps = torch.from_numpy(eps)
    model.eval()
    model.mode = "predicting"
    model.reparameterize = lambda mu, logvar: teps * torch.exp(0.5 * logvar) + mu          # models.py:76-79 with a fixed eps
    dummy = torch.zeros(B, 3, S)                           # p_input: only its shape is read in this mode (:440-452)
    with torch.no_grad():
        timg = torch.from_numpy(images)
        img_features = model.img_encoder(timg)
        out = model(dummy, dummy, timg)
    final = out["p_prior_samples"][-1]
    std = float(final.std())
    tgt = MO.svr_target(seed, B, S, std)
    r = final.transpose(1, 2).contiguous()
    t = torch.from_numpy(tgt).transpose(1, 2).contiguous()
    dl, dr = ref_utils.distChamferCUDA(r, t)                                                 # evaluating.py:201
    cd = dl.mean(1) + dr.mean(1)                                                             # :202 (before its .mean())
    f1 = ref_utils.f_score(r, t)                                                             # :203
    gold = {"seed": np.array(seed), "B": np.array(B), "S": np.array(S), "target_std": np.array(std, np.float64),
            "img_features": img_features.numpy(), "cd_per_cloud": cd.numpy(), "f_score": f1.numpy(),
            "final_first": np.ascontiguousarray(final[:keep_clouds].numpy()),
            "final_abs_sum": final.abs().sum((1, 2)).double().numpy(), "final_sum": final.sum((1, 2)).double().numpy(),
            "sum_p_logvars_first": sum(out["p_prior_logvars"][1:])[:keep_clouds].numpy(),
            "p_base_logvar": out["p_prior_logvars"][0][:, :, 0].numpy()}
    for k in ("g_prior_samples", "g_prior_mus", "g_prior_logvars"):
        gold[k + "_len"] = np.array(len(out[k]))
        for i in (0, 1, len(out[k]) - 1):
            gold["%s/%d" % (k, i)] = np.ascontiguousarray(out[k][i].numpy())
    gold["p_prior_samples_len"] = np.array(len(out["p_prior_samples"]))
    print("predicting golden (B=%d, S=%d, G=%d, %d layers): CD %.6f, F1 %.3f (min %.2f max %.2f), cloud std %.4f"
          % (B, S, G, len(out["p_prior_samples"]) - 1, float(cd.mean()), float(f1.mean()), float(f1.min()), float(f1.max()), std))
    return gold


def svr_sections(ref_models, ref_losses, mir_models, mir_losses, em, write):
    """the single-view-reconstruction model, Local_Cond_RNVP_MC_Global_RNVP_VAE_IC (models.py:261-464)."""
    from oracle import detrng
    ref_utils = _reference_utils(em)
    # ---- 4. training mode (models.py:326-371), reference vs the same model on the mirror's classes, CPU ---------------------
    cfg = dict(MO.SVR_CONFIG_SMALL, util_mode="training")
    G = cfg["g_latent_space_size"]
    state = MO.make_svr_state(31, cfg)
    B, N = 4, 96
    x, _, _ = MO.svr_inputs(31, B, N, G)
    images = detrng.normal_f32(detrng.key(31, "images"), (B, 4, 64, 64), 0.0, 1.0)
    tx, timg = torch.from_numpy(x), torch.from_numpy(images)
    results, resnet_state = [], None
    for models, losses in ((ref_models, ref_losses), (mir_models, mir_losses)):
        model = _build_ic(models, cfg, state, resnet_state)
        resnet_state = {k: v.clone() for k, v in model.img_encoder.state_dict().items()}
        loss_fn = losses.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)
        model.train()
        torch.manual_seed(5)
        out = model(tx, tx, timg)
        terms = loss_fn(tx, tx, out)
        terms[0].backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        bufs = {k: b.clone() for k, b in model.named_buffers()}
        results.append((out, terms, grads, bufs, type(model.pc_decoder).__module__))
    (o1, l1, g1, b1, m1), (o2, l2, g2, b2, m2) = results
    assert m1.startswith("libref") and m2.startswith("dpf_nets_amd"), (m1, m2)
    assert set(o1) == set(o2)
    worst = 0.0
    for k in sorted(o1):
        a, b = flat(o1[k]), flat(o2[k])
        assert len(a) == len(b), (k, len(a), len(b))
        for i, (u, v) in enumerate(zip(a, b)):
            err = float((u.detach() - v.detach()).abs().max() / (u.detach().abs().max() + 1e-30))
            worst = max(worst, err)
            assert u.shape == v.shape and err < 2e-5, ("svr output", k, i, err)
    for u, v, name in zip(l1, l2, ("loss", "pnll", "gnll", "gent")):
        assert abs(float(u) - float(v)) <= 2e-5 * max(1.0, abs(float(u))), (name, float(u), float(v))
    assert set(g1) == set(g2)
    # relative to the tensor's own largest entry, floored at 1e-5 of the largest gradient entry of the model (a BatchNorm bias
    # in front of another BatchNorm has a gradient of pure rounding noise: ~1e-6 here)
    gmax = max(float(v.abs().max()) for v in g1.values())
    gerr = {k: float((g1[k] - g2[k]).abs().max() / max(float(g1[k].abs().max()), 1e-5 * gmax)) for k in g1}
    bad = sorted(((e, k, float(g1[k].abs().max())) for k, e in gerr.items() if e >= 2e-3), reverse=True)
    assert not bad, ("svr grad", bad[:8])
    for k in b1:
        assert torch.allclose(b1[k].float(), b2[k].float(), rtol=1e-4, atol=1e-6), ("svr buffer", k)
    print("SVR model, training mode: %d output entries, 4 loss terms, %d gradients, %d buffers agree (worst output rel err %.2e)"
          % (len(o1), len(g1), len(b1), worst))

    # ---- 5. predicting mode (models.py:417-462 + evaluating.py:198-205), reference vs mirror classes, then the goldens -------
    cfg_p = dict(MO.SVR_CONFIG_SMALL, util_mode="predicting")
    st_p = MO.make_svr_state(37, cfg_p)
    mref = _build_ic(ref_models, cfg_p, st_p)
    gold_small = _predict_golden(mref, ref_utils, cfg_p, 37, 3, 200, 64, keep_clouds=3)
    mmir = _build_ic(mir_models, cfg_p, st_p, {k: v.clone() for k, v in mref.img_encoder.state_dict().items()})
    if torch.cuda.is_available():                  # the mirror's eval-mode blocks are HIP-only: compared on the GPU box
        pass
    write_or_check(gold_small, os.path.join(ROOT, "tests", "golden", "model_svr_small.npz"), write, "SVR predicting (small)")
    del mmir
    # the shipped shapes: configs/svr/all.yaml:6,10,62-88 -- B = 50 clouds of 2500 points, G = 512, 63 coupling layers
    cfg_f = dict(MO.SVR_CONFIG, util_mode="predicting")
    st_f = MO.make_svr_state(41, cfg_f)
    mfull = _build_ic(ref_models, cfg_f, st_f)
    gold_full = _predict_golden(mfull, ref_utils, cfg_f, 41, MO.SVR_BATCH, MO.SVR_CLOUD, 224, keep_clouds=6)
    write_or_check(gold_full, os.path.join(ROOT, "tests", "golden", "model_svr_predict.npz"), write, "SVR predicting (shipped shapes)")


def write_or_check(gold, path, write, what):
    if write:
        np.savez_compressed(path, **gold)
        print("wrote %s (%d arrays, %.0f KB)" % (path, len(gold), os.path.getsize(path) / 1024))
    else:
        old = np.load(path)
        for k in gold:
            if gold[k].dtype.kind in "US":
                assert list(old[k]) == list(gold[k]), k
            else:
                np.testing.assert_allclose(old[k], gold[k], rtol=1e-6, atol=1e-7, err_msg=k)
        print("%s: %s is up to date (%d arrays)" % (what, path, len(gold)))


if __name__ == "__main__":
    main()
