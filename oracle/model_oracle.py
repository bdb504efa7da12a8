"""Model-level test helpers: deterministic weights for a whole dpf-nets autoencoder and a restatement of the
EVALUATING-mode forward of `Local_Cond_RNVP_MC_Global_RNVP_VAE` over whatever building blocks it is given.

TEST INFRASTRUCTURE -- the checker, never the thing measured or shipped.  Only tests/, oracle/check_dropin.py,
bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it.

Parity status: PINNED.  oracle/check_dropin.py (build container only) runs the reference's own
lib/networks/models.py on these weights and writes tests/golden/model_eval.npz and model_train.npz;
tests/test_oracle_golden.py checks this file's `evaluating_forward` (over the CPU oracles) and tests/test_gpu_model.py the
HIP path (over the mirror classes of dpf_nets_amd.networks) against them -- `training_forward` + `vae_loss` with
loss.backward() being the whole training step of training.py:37-55.

  Local_Cond_RNVP_MC_Global_RNVP_VAE.__init__          lib/networks/models.py:13-74   (names of the sub-modules / parameters)
  ... .forward, mode == 'evaluating'                    lib/networks/models.py:173-216
  ... .forward, mode == 'training'                      lib/networks/models.py:125-171   (training_forward)
  Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss.forward       lib/networks/losses.py:18-51     (vae_loss)
  FeatureEncoder.forward (n_layers = 0|1, mus/logvars)  lib/networks/encoders.py:31-83
"""
import numpy as np
import torch

from . import detrng
from . import encoder_oracle as EO
from . import flow_oracle as FO
from . import gprior_oracle as GO

# a small instance of configs/autoencoding/all_scaled.yaml:27-51 (same structure: 'fixed' base distribution of
# airplane.yaml, G = 64, 2 prior flows, 2 decoder triples) -- small enough for a CPU reference run in seconds
CONFIG = dict(util_mode="evaluating", deterministic=False, pc_enc_init_n_channels=3, pc_enc_init_n_features=64,
              pc_enc_n_features=[128, 256, 512], g_latent_space_size=64, g_prior_n_flows=2, g_prior_n_features=32,
              g_posterior_n_layers=0, p_latent_space_size=3, p_prior_n_layers=0, p_decoder_n_flows=2,
              p_decoder_n_features=64, p_decoder_base_type="fixed", p_decoder_base_var=-3.5960,
              pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0)


def make_model_state(seed, cfg=CONFIG):
    """numpy state dict of the whole model with the reference's parameter / buffer names (models.py:38-74)."""
    G = cfg["g_latent_space_size"]
    st = {}
    for k, v in EO.make_encoder_state(seed + 1, cfg["pc_enc_init_n_channels"], cfg["pc_enc_init_n_features"],
                                      tuple(cfg["pc_enc_n_features"])).items():
        st["pc_encoder." + k] = v
    for k, v in GO.make_gprior_state(seed + 2, cfg["g_prior_n_flows"], cfg["g_prior_n_features"], G).items():
        st["g_prior." + k] = v
    for k, v in FO.make_decoder_state(seed + 3, cfg["p_decoder_n_flows"], cfg["p_decoder_n_features"], G).items():
        st["pc_decoder." + k] = v
    C = cfg["pc_enc_n_features"][-1]
    assert cfg["g_posterior_n_layers"] == 0 and cfg["p_decoder_base_type"] == "fixed"
    st["g_posterior.mus.mu_mlp0.weight"] = detrng.normal_f32(detrng.key(seed, "gp.mu.w"), (G, C), 0.0, 0.05)
    st["g_posterior.mus.mu_mlp0.bias"] = detrng.normal_f32(detrng.key(seed, "gp.mu.b"), (G,), 0.0, 0.05)
    st["g_posterior.logvars.logvar_mlp0.weight"] = detrng.normal_f32(detrng.key(seed, "gp.lv.w"), (G, C), 0.0, 0.03)
    st["g_posterior.logvars.logvar_mlp0.bias"] = detrng.normal_f32(detrng.key(seed, "gp.lv.b"), (G,), 0.0, 0.03)
    st["g0_prior_mus"] = detrng.normal_f32(detrng.key(seed, "g0.mu"), (1, G), 0.0, 0.033)
    st["g0_prior_logvars"] = detrng.normal_f32(detrng.key(seed, "g0.lv"), (1, G), 0.0, 0.33)
    st["p_prior_mus"] = np.zeros((1, cfg["p_latent_space_size"], 1), np.float32)
    st["p_prior_logvar"] = np.full((1, cfg["p_latent_space_size"], 1), cfg["p_decoder_base_var"], np.float32)
    return st


def model_inputs(seed, B, N):
    """(B,3,N) input clouds (encoder and decoder see the same cloud in the autoencoder) and the base noise eps."""
    x = detrng.normal_f32(detrng.key(seed, "cloud"), (B, 3, N), 0.0, 0.25)
    eps = detrng.normal_f32(detrng.key(seed, "eps"), (B, 3, N), 0.0, 1.0)
    return x, eps


def evaluating_forward(blocks, st, g_input, eps, n_sampled_points=None):
    """models.py:173-216 over `blocks` = dict(pc_encoder=f(x)->features or (B,512) max, g_prior=f(g, mode)->3 lists,
    pc_decoder=f(p, g, mode)->3 lists); `st` = torch tensors of the non-block parameters; eps replaces
    torch.randn_like of reparameterize (models.py:76-79) so that the result does not depend on a generator stream."""
    out = {}
    B = g_input.shape[0]
    S = g_input.shape[2] if n_sampled_points is None else n_sampled_points
    feats = blocks["pc_encoder"](g_input)
    g_enc = torch.max(feats, dim=2)[0]                                                       # :175
    out["g_posterior_mus"] = g_enc @ st["g_posterior.mus.mu_mlp0.weight"].t() + st["g_posterior.mus.mu_mlp0.bias"]
    out["g_posterior_logvars"] = g_enc @ st["g_posterior.logvars.logvar_mlp0.weight"].t() + \
        st["g_posterior.logvars.logvar_mlp0.bias"]
    out["g_posterior_samples"] = out["g_posterior_mus"]                                      # :178
    G = out["g_posterior_mus"].shape[1]
    out["g_prior_mus"] = [st["g0_prior_mus"].expand(B, G)]
    out["g_prior_logvars"] = [st["g0_prior_logvars"].expand(B, G)]
    buf_g = blocks["g_prior"](out["g_posterior_samples"], "inverse")                         # :182
    out["g_prior_samples"] = list(buf_g[0]) + [out["g_posterior_samples"]]
    out["g_prior_mus"] += list(buf_g[1])
    out["g_prior_logvars"] += list(buf_g[2])
    out["p_prior_mus"] = [st["p_prior_mus"].expand(B, 3, S)]                                 # :203-209 ('fixed')
    out["p_prior_logvars"] = [st["p_prior_logvar"].expand(B, 3, S)]
    z = eps * torch.exp(0.5 * out["p_prior_logvars"][0]) + out["p_prior_mus"][0]             # :211 / :76-79
    out["p_prior_samples"] = [z]
    buf = blocks["pc_decoder"](z.contiguous(), out["g_posterior_samples"], "direct")         # :213
    out["p_prior_samples"] += list(buf[0])
    out["p_prior_mus"] += list(buf[1])
    out["p_prior_logvars"] += list(buf[2])
    return out


def training_forward(blocks, st, g_input, p_input, eps_g):
    """models.py:125-171 ('fixed' base distribution, g_posterior_n_layers = 0) over `blocks` (as evaluating_forward; the
    modules are in train() mode) and `st` = the non-block parameters (torch tensors, requires_grad where the caller wants
    their gradients); eps_g (B,G) replaces torch.randn_like of reparameterize (models.py:76-79)."""
    out = {}
    B = g_input.shape[0]
    feats = blocks["pc_encoder"](g_input)                                                    # :130
    g_enc = torch.max(feats, dim=2)[0]                                                       # :131
    out["g_posterior_mus"] = torch.nn.functional.linear(g_enc, st["g_posterior.mus.mu_mlp0.weight"],
                                                        st["g_posterior.mus.mu_mlp0.bias"])               # :133
    out["g_posterior_logvars"] = torch.nn.functional.linear(g_enc, st["g_posterior.logvars.logvar_mlp0.weight"],
                                                            st["g_posterior.logvars.logvar_mlp0.bias"])
    out["g_posterior_samples"] = eps_g * torch.exp(0.5 * out["g_posterior_logvars"]) + out["g_posterior_mus"]   # :134 / :76-79
    G = out["g_posterior_mus"].shape[1]
    out["g_prior_mus"] = [st["g0_prior_mus"].expand(B, G)]                                   # :136-137
    out["g_prior_logvars"] = [st["g0_prior_logvars"].expand(B, G)]
    buf_g = blocks["g_prior"](out["g_posterior_samples"], "inverse")                         # :138
    out["g_prior_samples"] = list(buf_g[0]) + [out["g_posterior_samples"]]
    out["g_prior_mus"] += list(buf_g[1])
    out["g_prior_logvars"] += list(buf_g[2])
    N = p_input.shape[2]
    out["p_prior_mus"] = [st["p_prior_mus"].expand(B, 3, N)]                                 # :160-166 ('fixed')
    out["p_prior_logvars"] = [st["p_prior_logvar"].expand(B, 3, N)]
    buf_p = blocks["pc_decoder"](p_input, out["g_posterior_samples"], "inverse")             # :168
    out["p_prior_samples"] = buf_p[0] + [p_input]                                            # list-likes: the caller's own `+`
    out["p_prior_mus"] += buf_p[1]
    out["p_prior_logvars"] += buf_p[2]
    return out


def vae_loss(out, pnll_fn, cfg=CONFIG):
    """losses.py:37-51: (loss, pnll, gnll, gent); pnll_fn = the PointFlowNLL under test (losses.py:7-15)."""
    import math
    pnll = pnll_fn(out["p_prior_samples"], out["p_prior_mus"], out["p_prior_logvars"])
    s, m, lv = out["g_prior_samples"], out["g_prior_mus"], out["g_prior_logvars"]
    gnll = 0.5 * torch.add(torch.sum(sum(lv) + ((s[0] - m[0]) ** 2 / torch.exp(lv[0]))) / s[0].shape[0],
                           math.log(2.0 * math.pi) * s[0].shape[1])                          # GaussianFlowNLL :18-26
    plv = out["g_posterior_logvars"]
    gent = 0.5 * torch.add(plv.shape[1] * (1.0 + math.log(2.0 * math.pi)), plv.sum(1).mean())   # GaussianEntropy :29-34
    loss = cfg["pnll_weight"] * pnll + cfg["gnll_weight"] * gnll - cfg["gent_weight"] * gent
    return loss, pnll, gnll, gent
