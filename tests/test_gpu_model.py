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
