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


# ---------------------------------------------------------------------------------------------------------------------
# Single-view reconstruction model: Local_Cond_RNVP_MC_Global_RNVP_VAE_IC (lib/networks/models.py:261-464).
# The ResNet image encoder (models.py:287, resnet.py) is OUT of the hot path and stays the reference's own PyTorch code:
# the fixtures carry its output `img_features` (B, G), computed by the reference in the build container, and everything
# downstream of it -- g0_prior, the latent prior flow, p_prior, the point decoder, Chamfer and f_score -- is what the tests
# reproduce.
#   ... .forward, mode == 'predicting'                   lib/networks/models.py:417-462   (predicting_forward)
#   ... .forward, mode == 'training'                     lib/networks/models.py:326-371   (svr_training_forward)
#   evaluate(), predicting branch                        lib/networks/evaluating.py:198-205, utils.py:38-42
# ---------------------------------------------------------------------------------------------------------------------
# configs/svr/all.yaml:62-88 at reduced width (same structure: one hidden layer in every FeatureEncoder, 'freevar' base)
SVR_CONFIG_SMALL = dict(util_mode="predicting", deterministic=False, pc_enc_init_n_channels=3, pc_enc_init_n_features=64,
                        pc_enc_n_features=[128, 256, 512], g_latent_space_size=64, g_prior_n_layers=1, g_prior_n_flows=2,
                        g_prior_n_features=32, g_posterior_n_layers=1, p_latent_space_size=3, p_prior_n_layers=1,
                        p_decoder_n_flows=2, p_decoder_n_features=64, p_decoder_base_type="freevar", p_decoder_base_var=0.0,
                        pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0)
# configs/svr/all.yaml:62-88 as shipped (G = 512, 7 prior flows of 128 features, 21 decoder triples = 63 coupling layers)
# (`oracle_final_std`: read by make_svr_state only -- the std of the decoder's last SharedDots; 0.01 is the reference's own
# weight_std (decoders.py:42), which keeps a 63-layer stack of deterministic weights near the data's scale)
SVR_CONFIG = dict(SVR_CONFIG_SMALL, g_latent_space_size=512, g_prior_n_flows=7, g_prior_n_features=128, p_decoder_n_flows=21,
                  oracle_final_std=0.01)
SVR_BATCH, SVR_CLOUD = 50, 2500               # configs/svr/all.yaml:10, :6


def _feature_encoder_state(seed, name, n_layers, C, L, deterministic, mu_std, lv_std):
    """FeatureEncoder (encoders.py:31-83) with the reference's sub-module names; BatchNorm running statistics non-trivial."""
    st = {}
    for i in range(n_layers):
        p = "%s.features.mlp%d" % (name, i)
        st[p + ".weight"] = detrng.normal_f32(detrng.key(seed, p + ".w"), (C, C), 0.0, 1.0 / np.sqrt(C))
        st[p + "_bn.weight"] = detrng.uniform_f32(detrng.key(seed, p + ".g"), (C,), 0.7, 1.3)
        st[p + "_bn.bias"] = detrng.normal_f32(detrng.key(seed, p + ".b"), (C,), 0.0, 0.1)
        st[p + "_bn.running_mean"] = detrng.normal_f32(detrng.key(seed, p + ".rm"), (C,), 0.0, 0.1)
        st[p + "_bn.running_var"] = detrng.uniform_f32(detrng.key(seed, p + ".rv"), (C,), 0.5, 1.5)
        st[p + "_bn.num_batches_tracked"] = np.array(3, np.int64)
    st[name + ".mus.mu_mlp0.weight"] = detrng.normal_f32(detrng.key(seed, name + ".mu.w"), (L, C), 0.0, mu_std)
    st[name + ".mus.mu_mlp0.bias"] = detrng.normal_f32(detrng.key(seed, name + ".mu.b"), (L,), 0.0, mu_std)
    if not deterministic:
        st[name + ".logvars.logvar_mlp0.weight"] = detrng.normal_f32(detrng.key(seed, name + ".lv.w"), (L, C), 0.0, lv_std)
        st[name + ".logvars.logvar_mlp0.bias"] = detrng.normal_f32(detrng.key(seed, name + ".lv.b"), (L,), 0.0, lv_std)
    return st


def make_svr_state(seed, cfg=SVR_CONFIG_SMALL):
    """numpy state dict of the IC model WITHOUT its ResNet (models.py:287-320): reference parameter / buffer names."""
    G, C = cfg["g_latent_space_size"], cfg["pc_enc_n_features"][-1]
    assert cfg["p_decoder_base_type"] == "freevar"
    st = {}
    for k, v in EO.make_encoder_state(seed + 1, cfg["pc_enc_init_n_channels"], cfg["pc_enc_init_n_features"],
                                      tuple(cfg["pc_enc_n_features"])).items():
        st["pc_encoder." + k] = v
    for k, v in GO.make_gprior_state(seed + 2, cfg["g_prior_n_flows"], cfg["g_prior_n_features"], G).items():
        st["g_prior." + k] = v
    kw = {"final_std": cfg["oracle_final_std"]} if "oracle_final_std" in cfg else {}
    for k, v in FO.make_decoder_state(seed + 3, cfg["p_decoder_n_flows"], cfg["p_decoder_n_features"], G, **kw).items():
        st["pc_decoder." + k] = v
    st.update(_feature_encoder_state(seed + 4, "g0_prior", cfg["g_prior_n_layers"], G, G, False, 0.05, 0.03))
    st.update(_feature_encoder_state(seed + 5, "g_posterior", cfg["g_posterior_n_layers"], C, G, False, 0.05, 0.03))
    st.update(_feature_encoder_state(seed + 6, "p_prior", cfg["p_prior_n_layers"], G, cfg["p_latent_space_size"], True, 0.05, 0.0))
    # the 'freevar' base distribution's log-variance head around the autoencoder configs' fixed value (-3.5960,
    # configs/autoencoding/all_scaled.yaml:51): sampled clouds then have the extent at which nearest-neighbour distances
    # straddle f_score's 0.001 threshold
    st["p_prior.mus.mu_mlp0.bias"] = (st["p_prior.mus.mu_mlp0.bias"] - np.float32(3.596)).astype(np.float32)
    st["p_prior_mus"] = np.zeros((1, cfg["p_latent_space_size"], 1), np.float32)
    return st


def svr_inputs(seed, B, N, G, S=None):
    """clouds (B,3,N) for the encoder / the training decoder, base noise eps (B,3,S) for the sampled cloud, the noise of
    the latent reparameterisation eps_g (B,G).  Images are made by oracle/check_dropin.py (they only feed the ResNet)."""
    S = N if S is None else S
    x = detrng.normal_f32(detrng.key(seed, "cloud"), (B, 3, N), 0.0, 0.25)
    eps = detrng.normal_f32(detrng.key(seed, "eps"), (B, 3, S), 0.0, 1.0)
    eps_g = detrng.normal_f32(detrng.key(seed, "eps_g"), (B, G), 0.0, 1.0)
    return x, eps, eps_g


def feature_encoder(st, name, x, n_layers, deterministic, training=False, stats_out=None):
    """FeatureEncoder.forward (encoders.py:75-83) on state tensors: n_layers x [Linear . BatchNorm1d . Swish], heads."""
    f = x
    for i in range(n_layers):
        p = "%s.features.mlp%d" % (name, i)
        f = torch.nn.functional.linear(f, st[p + ".weight"])
        if training:
            var, mean = torch.var_mean(f, dim=0, unbiased=False)
            if stats_out is not None:
                stats_out[p + "_bn"] = (mean.detach(), var.detach() * (f.shape[0] / (f.shape[0] - 1.0)))
        else:
            mean, var = st[p + "_bn.running_mean"], st[p + "_bn.running_var"]
        f = (f - mean) / torch.sqrt(var + 1e-5) * st[p + "_bn.weight"] + st[p + "_bn.bias"]
        f = f * torch.sigmoid(f)                                                              # Swish, layers.py:9-10
    mu = torch.nn.functional.linear(f, st[name + ".mus.mu_mlp0.weight"], st[name + ".mus.mu_mlp0.bias"])
    if deterministic:
        return mu
    return mu, torch.nn.functional.linear(f, st[name + ".logvars.logvar_mlp0.weight"], st[name + ".logvars.logvar_mlp0.bias"])


def predicting_forward(blocks, img_features, eps, cfg=SVR_CONFIG_SMALL):
    """models.py:417-462 ('freevar' base) over `blocks` = dict(g0_prior=f(x)->(mu, logvar), g_prior=f(g, mode)->3 lists,
    p_prior=f(g)->(B,3), pc_decoder=f(p, g, mode)->3 lists) and `p_prior_mus` (1,3,1); img_features (B,G) is the ResNet's
    output (models.py:418); eps (B,3,S) replaces torch.randn_like of reparameterize (:76-79 / :456)."""
    out = {}
    B, S = eps.shape[0], eps.shape[2]
    mu0, lv0 = blocks["g0_prior"](img_features)                                              # :420
    out["g_prior_mus"], out["g_prior_logvars"] = [mu0], [lv0]
    out["g_prior_samples"] = [mu0]                                                           # :422
    buf_g = blocks["g_prior"](mu0, "direct")                                                 # :423
    out["g_prior_samples"] += list(buf_g[0])
    out["g_prior_mus"] += list(buf_g[1])
    out["g_prior_logvars"] += list(buf_g[2])
    g = out["g_prior_samples"][-1]
    out["p_prior_mus"] = [blocks["p_prior_mus"].expand(B, 3, S)]                             # :438-441 ('freevar')
    out["p_prior_logvars"] = [blocks["p_prior"](g).unsqueeze(2).expand(B, 3, S)]             # :442-444
    z = eps * torch.exp(0.5 * out["p_prior_logvars"][0]) + out["p_prior_mus"][0]             # :456
    out["p_prior_samples"] = [z]
    buf_p = blocks["pc_decoder"](z.contiguous(), g, "direct")                                # :458
    out["p_prior_samples"] += list(buf_p[0])
    out["p_prior_mus"] += list(buf_p[1])
    out["p_prior_logvars"] += list(buf_p[2])
    return out


def svr_target(seed, B, S, std):
    """Ground-truth clouds of the predicting-mode evaluation fixture: N(0, std) points, (B,3,S) -- independent of the model, at
    the density where nearest-neighbour distances straddle f_score's 0.001 threshold (utils.py:38)."""
    return detrng.normal_f32(detrng.key(seed, "target"), (B, 3, S), 0.0, std)
