"""Model level: the EVALUATING-mode forward of the reference's autoencoder (lib/networks/models.py:173-216) assembled from
the mirror classes only (dpf_nets_amd.networks: fused PointNet encoder, one-launch latent prior flow, fused point decoder,
HIP Chamfer, HIP PointFlowNLL) against the golden the reference's own model produced on CPU
(oracle/check_dropin.py -> tests/golden/model_eval.npz).  The list handling is the caller's: `+=` of a python list with the
decoder's list-likes, `[-1]` / `[0]` indexing, `sum()` (models.py:183-186, 211-216; evaluating.py:85, 110-113)."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle import model_oracle as MO

pytestmark = pytest.mark.gpu

# elementwise-relative error the whole model's parameter gradients are held to through their (sum, random projection, 1-norm)
# goldens (tests/gradcheck.py): decoder f16x3 with fp16 hi/lo gradient operands, encoder / prior flow at their own split
# precisions, one training step of the reference's model
MODEL_GRAD_TOL = 1e-3


def _build(nets, cfg, st):
    dev = torch.device("cuda", 0)
    enc = nets.PointNetCloudEncoder(cfg["pc_enc_init_n_channels"], cfg["pc_enc_init_n_features"], cfg["pc_enc_n_features"])
    enc.load_state_dict(FO.sub_state(st, "pc_encoder."), strict=True)
    prior = nets.GlobalRNVPDecoder(cfg["g_prior_n_flows"], cfg["g_prior_n_features"], cfg["g_latent_space_size"])
    prior.load_state_dict(FO.sub_state(st, "g_prior."), strict=True)
    dec = nets.LocalCondRNVPDecoder(cfg["p_decoder_n_flows"], cfg["p_decoder_n_features"], cfg["g_latent_space_size"])
    dec.load_state_dict(FO.sub_state(st, "pc_decoder."), strict=True)
    return enc.to(dev).eval(), prior.to(dev).eval(), dec.to(dev).eval()


def test_evaluating_forward_through_hip_vs_reference_model_golden(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.utils import distChamferCUDA
    gold = np.load(os.path.join(golden_dir, "model_eval.npz"))
    cfg = MO.CONFIG
    st = FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg))
    enc, prior, dec = _build(nets, cfg, st)
    dev = torch.device("cuda", 0)
    gst = {k: v.to(dev) for k, v in st.items() if not k.startswith(("pc_encoder.", "g_prior.", "pc_decoder."))}
    x = torch.from_numpy(gold["x"]).to(dev)
    eps = torch.from_numpy(gold["eps"]).to(dev)
    blocks = {"pc_encoder": enc, "g_prior": lambda g, mode: prior(g, mode=mode), "pc_decoder": lambda p, g, mode: dec(p, g, mode=mode)}
    with torch.no_grad():
        out = MO.evaluating_forward(blocks, gst, x, eps)

    def close(got, ref, what, rtol=1e-4, atol_scale=5e-6):
        ref = np.asarray(ref)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=rtol, atol=atol_scale * max(1.0, float(np.abs(ref).max())), err_msg=what)
    close(out["g_posterior_mus"], gold["g_posterior_mus"], "g_posterior_mus", atol_scale=2e-5)       # encoder: bf16x3 (1e-5 class)
    for k in ("g_prior_samples", "g_prior_logvars", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
        assert len(out[k]) == int(gold[k + "_len"]), k
        for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
            close(out[k][i], gold["%s/%d" % (k, i)], "%s/%d" % (k, i), atol_scale=3e-5)
    close(sum(out["p_prior_logvars"]), gold["sum_p_logvars"], "sum(p_prior_logvars)", atol_scale=3e-5)
    # PointFlowNLL exactly as losses.py:48 calls it (the HIP evaluation path: stride-0 base tensors, kernel's layer sum)
    pnll = nets.PointFlowNLL()(out["p_prior_samples"], out["p_prior_mus"], out["p_prior_logvars"])
    np.testing.assert_allclose(float(pnll), float(gold["pnll_as_losses_py"]), rtol=2e-5)
    # reconstruction CD as evaluating.py:85,110-113
    r = out["p_prior_samples"][-1].transpose(1, 2).contiguous()
    dl, dr = distChamferCUDA(r, x.transpose(1, 2).contiguous())
    cd = (dl.mean(1) + dr.mean(1))
    np.testing.assert_allclose(cd.cpu().numpy(), gold["cd_per_cloud"], rtol=2e-4)


@pytest.mark.parametrize("which", ["small", "predict"])
def test_svr_predicting_forward_cd_fscore_through_hip_vs_reference_model_golden(golden_dir, which):
    """the `predicting` forward of the reference's single-view-reconstruction model
    (Local_Cond_RNVP_MC_Global_RNVP_VAE_IC, models.py:417-462) downstream of its ResNet, assembled from the mirror classes
    only -- FeatureEncoder heads, the one-launch latent prior flow (direct), the fused point decoder (direct, 'freevar' base) --
    and the evaluation of evaluating.py:198-205: HIP Chamfer, per-cloud CD, f_score (utils.py:38-42).  `predict` = the SHIPPED
    shapes of configs/svr/all.yaml:6,10,62-88: B = 50 clouds of 2500 points, G = 512, 7 prior flows, 63 coupling layers.
    Golden: what the reference's own model + utils produced on CPU (oracle/check_dropin.py -> tests/golden/model_svr_*.npz;
    the fixture carries the ResNet's output, the ResNet itself stays the reference's PyTorch code)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd.networks.utils import distChamferCUDA, f_score
    gold = np.load(os.path.join(golden_dir, "model_svr_%s.npz" % which))
    cfg = MO.SVR_CONFIG_SMALL if which == "small" else MO.SVR_CONFIG
    seed, B, S = int(gold["seed"]), int(gold["B"]), int(gold["S"])
    G = cfg["g_latent_space_size"]
    if which == "predict":
        assert (B, S, G, cfg["p_decoder_n_flows"]) == (50, 2500, 512, 21)
    st = FO.to_torch(MO.make_svr_state(seed, cfg))
    dev = torch.device("cuda", 0)
    g0 = nets.FeatureEncoder(cfg["g_prior_n_layers"], G, G, deterministic=False)
    g0.load_state_dict(FO.sub_state(st, "g0_prior."), strict=True)
    pp = nets.FeatureEncoder(cfg["p_prior_n_layers"], G, cfg["p_latent_space_size"], deterministic=True)
    pp.load_state_dict(FO.sub_state(st, "p_prior."), strict=True)
    prior = nets.GlobalRNVPDecoder(cfg["g_prior_n_flows"], cfg["g_prior_n_features"], G)
    prior.load_state_dict(FO.sub_state(st, "g_prior."), strict=True)
    dec = nets.LocalCondRNVPDecoder(cfg["p_decoder_n_flows"], cfg["p_decoder_n_features"], G)
    dec.load_state_dict(FO.sub_state(st, "pc_decoder."), strict=True)
    g0, pp, prior, dec = (m.to(dev).eval() for m in (g0, pp, prior, dec))
    _, eps, _ = MO.svr_inputs(seed, B, S, G)
    blocks = {"g0_prior": g0, "p_prior": pp, "p_prior_mus": st["p_prior_mus"].to(dev),
              "g_prior": lambda g, mode: prior(g, mode=mode), "pc_decoder": lambda p, g, mode: dec(p, g, mode=mode)}
    with torch.no_grad():
        out = MO.predicting_forward(blocks, torch.from_numpy(gold["img_features"]).to(dev), torch.from_numpy(eps).to(dev), cfg)

    def close(got, ref, what, rtol=1e-4, atol_scale=3e-5):
        ref = np.asarray(ref)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=rtol, atol=atol_scale * max(1e-3, float(np.abs(ref).max())), err_msg=what)
    for k in ("g_prior_samples", "g_prior_mus", "g_prior_logvars"):
        assert len(out[k]) == int(gold[k + "_len"]), k
        for i in (0, 1, len(out[k]) - 1):
            close(out[k][i], gold["%s/%d" % (k, i)], "%s/%d" % (k, i))
    assert len(out["p_prior_samples"]) == int(gold["p_prior_samples_len"])
    close(out["p_prior_logvars"][0][:, :, 0], gold["p_base_logvar"], "base logvar")
    final = out["p_prior_samples"][-1]
    nf = gold["final_first"].shape[0]
    close(final[:nf], gold["final_first"], "final clouds")                                    # north star: 1e-4 rel
    close(sum(out["p_prior_logvars"][1:])[:nf], gold["sum_p_logvars_first"], "sum of the layers' logvars")
    np.testing.assert_allclose(final.abs().sum((1, 2)).double().cpu().numpy(), gold["final_abs_sum"], rtol=2e-5)
    np.testing.assert_allclose(final.sum((1, 2)).double().cpu().numpy(), gold["final_sum"], rtol=1e-4,
                               atol=2e-5 * float(gold["final_abs_sum"].max()))
    # ---- evaluating.py:198-205: transpose, distChamferCUDA, cd, f_score
    tgt = torch.from_numpy(MO.svr_target(seed, B, S, float(gold["target_std"]))).to(dev)
    r, t = final.transpose(1, 2).contiguous(), tgt.transpose(1, 2).contiguous()
    dl, dr = distChamferCUDA(r, t)
    cd = dl.mean(1) + dr.mean(1)
    np.testing.assert_allclose(cd.cpu().numpy(), gold["cd_per_cloud"], rtol=2e-4)
    f1 = f_score(r, t).cpu().numpy()
    # a distance within rounding of the 0.001 threshold may fall either side (the reference's distChamfer is the expanded
    # form, and the clouds themselves agree to 1e-4): a flipped point moves precision or recall by 100 / S
    assert np.all(np.abs(f1 - gold["f_score"]) <= 3 * 100.0 / S + 1e-3 * gold["f_score"]), np.abs(f1 - gold["f_score"]).max()
    assert abs(float(f1.mean()) - float(gold["f_score"].mean())) <= 100.0 / S


def test_training_forward_backward_through_hip_vs_reference_model_golden(golden_dir):
    """VERDICT r02 missing #1: ONE training step of the whole autoencoder (training.py:37-55 = models.py:125-171, the four
    loss terms of losses.py:37-51, loss.backward()) with every block on its HIP TRAINING kernels -- PointNet encoder
    (csrc/encoder_train.hip), latent prior flow (csrc/gprior_train.hip), point decoder (csrc/flow_train.hip, flattened
    store), PointFlowNLL (csrc/nll.hip) -- assembled from the mirror classes only, against what the reference's own
    model produced on CPU (oracle/check_dropin.py -> tests/golden/model_train.npz): outputs, loss terms, projections of all
    250 parameter gradients, BatchNorm running statistics after the step.  Run six times on the same inputs: the first
    calls launch eagerly, the later ones are served by hipGraph replays once the allocator hands the same blocks back
    (csrc/graph_cache.h), and every call must reproduce the same golden."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd._lib import lib
    from oracle.gen_golden import _grad_projection
    gold = np.load(os.path.join(golden_dir, "model_train.npz"))
    cfg = MO.CONFIG
    st = FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg))
    enc, prior, dec = _build(nets, cfg, st)
    enc.train(); prior.train(); dec.train()
    enc.hip_training = True
    dec.flatten_parameters()
    prior.flatten_parameters()
    dev = torch.device("cuda", 0)
    gst = {k: v.to(dev) for k, v in st.items() if not k.startswith(("pc_encoder.", "g_prior.", "pc_decoder."))}
    leaf = [k for k in gst if k.startswith(("g_posterior.", "g0_prior"))]       # nn.Parameters of the reference model
    for k in leaf:
        gst[k].requires_grad_(True)
    x = torch.from_numpy(gold["x"]).to(dev)
    eps_g = torch.from_numpy(gold["eps_g"]).to(dev)
    blocks = {"pc_encoder": enc, "g_prior": lambda g, mode: prior(g, mode=mode), "pc_decoder": lambda p, g, mode: dec(p, g, mode=mode)}
    pnll_fn = nets.PointFlowNLL()
    names = [str(k) for k in gold["grad_names"]]

    def close(got, ref, what, rtol=1e-4, atol_scale=1e-4):
        ref = np.asarray(ref)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=rtol, atol=atol_scale * max(1e-3, float(np.abs(ref).max())),
                                   err_msg=what)

    def named_grads():
        out = {}
        for pre, mod in (("pc_encoder.", enc), ("g_prior.", prior), ("pc_decoder.", dec)):
            for k, p in mod.named_parameters():
                out[pre + k] = p.grad
        for k in leaf:
            out[k] = gst[k].grad
        return out

    replays0 = int(lib().dpf_train_graph_replays())
    for call in range(6):
        for mod in (enc, prior, dec):
            mod.zero_grad(set_to_none=True)
        for k in leaf:
            gst[k].grad = None
        out = MO.training_forward(blocks, gst, x, x, eps_g)
        loss, pnll, gnll, gent = MO.vae_loss(out, pnll_fn, cfg)
        loss.backward()
        tag = "call %d: " % call
        # ---- the four loss terms (losses.py:51)
        got = np.array([float(loss.detach()), float(pnll.detach()), float(gnll.detach()), float(gent.detach())])
        np.testing.assert_allclose(got, gold["loss"], rtol=5e-5, err_msg=tag + "loss terms")
        # ---- outputs (north star: 1e-4 rel)
        for k in ("g_posterior_mus", "g_posterior_logvars", "g_posterior_samples"):
            close(out[k], gold[k], tag + k)
        for k in ("g_prior_samples", "p_prior_samples", "p_prior_mus", "p_prior_logvars"):
            assert len(out[k]) == int(gold[k + "_len"]), k
            for i in (0, 1, len(out[k]) // 2, len(out[k]) - 1):
                close(out[k][i], gold["%s/%d" % (k, i)], tag + "%s/%d" % (k, i))
        close(sum(out["p_prior_logvars"]), gold["sum_p_logvars"], tag + "sum(p_prior_logvars)")
        # ---- all 250 parameter gradients, two projections + the L1 norm each
        grads = named_grads()
        assert sorted(grads) == sorted(names) and all(g is not None for g in grads.values())
        worst = 0.0
        from tests.gradcheck import check_projections
        cpu_grads = {k: grads[k].detach().cpu() for k in names}
        check_projections(cpu_grads, _grad_projection([(k, cpu_grads[k]) for k in names], 23), lambda k: gold["gradproj/" + k],
                          MODEL_GRAD_TOL, tag)
        # ---- BatchNorm running statistics / counters after ONE step (the first call)
        if call == 0:
            nbuf = 0
            for pre, mod in (("pc_encoder.", enc), ("g_prior.", prior), ("pc_decoder.", dec)):
                for k, b in mod.named_buffers():
                    ref = gold["buffer/" + pre + k]
                    nbuf += 1
                    if ref.dtype.kind in "iu":
                        assert int(b) == int(ref), pre + k
                    else:
                        np.testing.assert_allclose(b.detach().cpu().numpy(), ref, rtol=2e-4, atol=2e-6, err_msg=pre + k)
            assert nbuf + 2 == sum(1 for k in gold.files if k.startswith("buffer/"))      # + p_prior_mus, p_prior_logvar
        del out, loss, pnll, gnll, gent, grads          # as a training loop: nothing of step k is alive in step k + 1
    assert int(lib().dpf_train_graph_replays()) - replays0 >= 2, "none of the later calls was served by graph replays"


@pytest.mark.parametrize("overlap", [True, False], ids=["prior_beside_decoder", "prior_in_line"])
def test_model_mirror_training_step_on_gpu_vs_reference_golden(golden_dir, overlap):
    """The mirror of the model class itself (networks/models.py: what `bench.py --leg train --model autoencoder` and the
    data-parallel worker run) on the GPU, every block on its HIP training kernels, the latent prior flow on a side stream
    BESIDE the point decoder (forward and backward) or in line: four steps against the reference model's golden training
    step (tests/golden/model_train.npz) -- loss terms, all 250 gradient projections -- through a GradArena, whose flat buffer
    must hold exactly those gradients after `sync()`; both schedules must give bit-identical gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks as nets, distributed as D
    from oracle.gen_golden import _grad_projection
    gold = np.load(os.path.join(golden_dir, "model_train.npz"))
    cfg = dict(MO.CONFIG, util_mode="training")
    dev = torch.device("cuda", 0)
    model = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg)
    model.load_state_dict(FO.to_torch(MO.make_model_state(int(gold["seed"]), cfg)), strict=True)
    model = model.to(dev).train()
    model.overlap_prior = overlap
    model.flatten_parameters()
    loss_fn = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)
    eps_g = torch.from_numpy(gold["eps_g"]).to(dev)
    model.reparameterize = lambda mu, logvar: eps_g * torch.exp(0.5 * logvar) + mu
    x = torch.from_numpy(gold["x"]).to(dev)
    arena = D.GradArena(model.parameters())
    names = [str(k) for k in gold["grad_names"]]
    assert [k for k, _ in model.named_parameters()] == names
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    bufs = []
    for call in range(4):
        model.load_state_dict(state0)                     # the same step again (running statistics rewound)
        arena.zero_grad()
        out = model(x, x)
        loss, pnll, gnll, gent = loss_fn(x, x, out)
        loss.backward()
        arena.allreduce()                                 # single process: sync() only
        got = np.array([float(v.detach()) for v in (loss, pnll, gnll, gent)])
        np.testing.assert_allclose(got, gold["loss"], rtol=5e-5, err_msg="call %d" % call)
        from tests.gradcheck import check_projections
        cpu_grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
        check_projections(cpu_grads, _grad_projection(list(cpu_grads.items()), 23), lambda k: gold["gradproj/" + k],
                          MODEL_GRAD_TOL, "call %d" % call)
        for k, p in model.named_parameters():             # every .grad lives in the ONE message buffer
            assert p.grad.untyped_storage().data_ptr() == arena.buf.untyped_storage().data_ptr(), k
        bufs.append(arena.buf.clone())
        del out, loss, pnll, gnll, gent
    for b in bufs[1:]:
        assert torch.equal(b, bufs[0])                    # deterministic, eager or replayed, whatever the streams' timing
    test_model_mirror_training_step_on_gpu_vs_reference_golden.bufs = getattr(
        test_model_mirror_training_step_on_gpu_vs_reference_golden, "bufs", {})
    test_model_mirror_training_step_on_gpu_vs_reference_golden.bufs[overlap] = bufs[0].cpu()
    both = test_model_mirror_training_step_on_gpu_vs_reference_golden.bufs
    if len(both) == 2:
        assert torch.equal(both[True], both[False])       # the side-stream schedule changes nothing but the timing


def test_model_mirror_steady_state_step_copies_nothing_from_the_host():
    """The whole autoencoder's training step (mirror classes, HIP blocks, GradArena, mirror Adam) issues no host-to-device copy,
    no advanced indexing and no scalar read-back once it is in steady state: any of them blocks the host behind the stream and
    starves the GPU (the r03 finding in the decoder's running-statistics update: 0.6 ms of a 4.4 ms step)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from torch.profiler import profile, ProfilerActivity
    from dpf_nets_amd import networks as nets, distributed as D
    cfg = dict(MO.CONFIG, util_mode="training")
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    model = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg).to(dev).train()
    model.flatten_parameters()
    loss_fn = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)
    arena = D.GradArena(model.parameters())
    opt = nets.Adam(list(model.parameters()), lr=1e-4, amsgrad=True)
    x = (torch.randn(6, 3, 640, generator=torch.Generator().manual_seed(1)) * 0.25).to(dev)

    def step():
        arena.zero_grad()
        loss_fn(x, x, model(x, x))[0].backward()
        arena.allreduce()
        opt.step()
    for _ in range(6):
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
