"""The autoencoder around the hot path: mirror of lib/networks/models.py:13-258 (`Local_Cond_RNVP_MC_Global_RNVP_VAE`).

The reference's own models.py runs unchanged on this package's classes (oracle/check_dropin.py proves it); this mirror
exists so that a whole training / evaluation step -- encoder -> posterior -> latent prior flow -> point decoder -> loss --
can be assembled, benchmarked and data-parallelised on a box that has only this package: every block on its HIP kernels,
every gradient of the model in ONE flat message (distributed.GradArena).  Same constructor kwargs, sub-module and
parameter names (reference checkpoints load with strict=True), same output dict, same list handling.
"""
import torch
import torch.nn as nn

from .decoders import LocalCondRNVPDecoder
from .encoders import FeatureEncoder, PointNetCloudEncoder
from .prior_flows import GlobalRNVPDecoder


class _JoinMain(torch.autograd.Function):
    """Identity placed at the input of a block that runs on a side stream.  Its backward runs on that side stream AFTER the
    block's backward (autograd runs a node's backward on the stream of its forward) and makes the main stream wait for it:
    whatever the block's backward wrote behind autograd's back (a flat store's gradient buffer) is ordered before the
    main stream's next work (gradient exchange, optimizer)."""

    @staticmethod
    def forward(ctx, x, main):
        ctx.main = main
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dx):
        ctx.main.wait_stream(torch.cuda.current_stream(dx.device))
        return dx, None


class Local_Cond_RNVP_MC_Global_RNVP_VAE(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        for k in ("deterministic", "pc_enc_init_n_channels", "pc_enc_init_n_features", "pc_enc_n_features",
                  "g_latent_space_size", "g_prior_n_flows", "g_prior_n_features", "g_posterior_n_layers",
                  "p_latent_space_size", "p_prior_n_layers", "p_decoder_n_flows", "p_decoder_n_features",
                  "p_decoder_base_type", "p_decoder_base_var"):
            setattr(self, k, kwargs.get(k))
        self.mode = kwargs.get("util_mode")
        G, P = self.g_latent_space_size, self.p_latent_space_size
        self.pc_encoder = PointNetCloudEncoder(self.pc_enc_init_n_channels, self.pc_enc_init_n_features, self.pc_enc_n_features)
        self.g0_prior_mus = nn.Parameter(torch.empty(1, G))                                     # models.py:42-46
        self.g0_prior_logvars = nn.Parameter(torch.empty(1, G))
        with torch.no_grad():
            nn.init.normal_(self.g0_prior_mus, mean=0.0, std=0.033)
            nn.init.normal_(self.g0_prior_logvars, mean=0.0, std=0.33)
        self.g_prior = GlobalRNVPDecoder(self.g_prior_n_flows, self.g_prior_n_features, G, weight_std=0.01)
        self.g_posterior = FeatureEncoder(self.g_posterior_n_layers, self.pc_enc_n_features[-1], G, deterministic=False,
                                          mu_weight_std=0.0033, mu_bias=0.0, logvar_weight_std=0.033, logvar_bias=0.0)
        if self.p_decoder_base_type == "free":                                                 # models.py:56-69
            self.p_prior = FeatureEncoder(self.p_prior_n_layers, G, P, deterministic=False, mu_weight_std=0.001, mu_bias=0.0,
                                          logvar_weight_std=0.01, logvar_bias=0.0)
        elif self.p_decoder_base_type == "freevar":
            self.register_buffer("p_prior_mus", torch.zeros((1, P, 1)))
            self.p_prior = FeatureEncoder(self.p_prior_n_layers, G, P, deterministic=True, mu_weight_std=0.01, mu_bias=0.0)
        elif self.p_decoder_base_type == "fixed":
            self.register_buffer("p_prior_mus", torch.zeros((1, P, 1)))
            self.register_buffer("p_prior_logvar", self.p_decoder_base_var * torch.ones((1, P, 1)))
        self.pc_decoder = LocalCondRNVPDecoder(self.p_decoder_n_flows, self.p_decoder_n_features, G, weight_std=0.01)
        # training steps on the GPU: the latent prior flow (14 steps of ~8 tiny dependent launches, forward and backward) and
        # the point decoder both depend only on the posterior sample, so the prior runs on a side stream BESIDE the decoder's
        # kernels, forward and -- because autograd runs a node's backward on its forward's stream -- backward (r03).  Measured:
        # 7.69 -> 7.42 ms per step at B=32, N=2048 (one round of decoder workgroups: the prior's small launches find room beside
        # them), 11.05 -> 11.21 ms at B=64 (two rounds: the chip is full and they only get in the way) -- hence the size rule.
        self.overlap_prior = True
        self.overlap_prior_max_points = 65536
        object.__setattr__(self, "_side_streams", {})

    def _prior_beside(self, g, run_rest):
        """g_prior(g, 'inverse') on a side stream while run_rest() (the point decoder) is issued on the current one."""
        cur = torch.cuda.current_stream(g.device)
        side = self._side_streams.get(g.device)
        if side is None:
            side = self._side_streams[g.device] = torch.cuda.Stream(g.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g.record_stream(side)
            buf_g = self.g_prior(_JoinMain.apply(g, cur), mode="inverse")
        rest = run_rest()
        cur.wait_stream(side)
        for lst in buf_g:
            for t in lst:
                t.record_stream(cur)
        return buf_g, rest

    def reparameterize(self, mu, logvar):                                                      # models.py:76-79
        std = torch.exp(0.5 * logvar)
        eps = torch.randn_like(std)
        return eps.mul(std).add_(mu)

    def _base(self, g, B, S):
        """The base distribution of the point flow given the cloud code g: ([mu (B,P,S)], [logvar (B,P,S)]), stride-0
        expansions as the reference builds them (models.py:140-166 and its copies at :187-209, :226-248, :93-115)."""
        P = self.p_latent_space_size
        if self.p_decoder_base_type == "free":
            mu, lv = self.p_prior(g)
            return [mu.unsqueeze(2).expand(B, P, S)], [lv.unsqueeze(2).expand(B, P, S)]
        if self.p_decoder_base_type == "freevar":
            return [self.p_prior_mus.expand(B, P, S)], [self.p_prior(g).unsqueeze(2).expand(B, P, S)]
        return [self.p_prior_mus.expand(B, P, S)], [self.p_prior_logvar.expand(B, P, S)]

    def encode(self, g_input):                                                                 # models.py:81-88
        g_enc = torch.max(self.pc_encoder(g_input), dim=2)[0]
        return {"g_posterior_mus": self.g_posterior(g_enc)[0]}

    def decode(self, g_sample, n_sampled_points=2048):                                         # models.py:90-123
        out = {}
        out["p_prior_mus"], out["p_prior_logvars"] = self._base(g_sample, g_sample.shape[0], n_sampled_points)
        out["p_prior_samples"] = [self.reparameterize(out["p_prior_mus"][0], out["p_prior_logvars"][0])]
        buf = self.pc_decoder(out["p_prior_samples"][0], g_sample, mode="direct")
        out["p_prior_samples"] += buf[0]
        out["p_prior_mus"] += buf[1]
        out["p_prior_logvars"] += buf[2]
        return out

    def forward(self, g_input, p_input, n_sampled_points=None):                                # models.py:125-258
        S = p_input.shape[2] if n_sampled_points is None else n_sampled_points
        B, G = g_input.shape[0], self.g_latent_space_size
        out = {}
        if self.mode in ("training", "evaluating"):
            g_enc = torch.max(self.pc_encoder(g_input), dim=2)[0]                              # :130-131 / :174-175
            out["g_posterior_mus"], out["g_posterior_logvars"] = self.g_posterior(g_enc)
            out["g_posterior_samples"] = self.reparameterize(out["g_posterior_mus"], out["g_posterior_logvars"]) \
                if self.mode == "training" else out["g_posterior_mus"]                         # :134 / :178
            out["g_prior_mus"] = [self.g0_prior_mus.expand(B, G)]
            out["g_prior_logvars"] = [self.g0_prior_logvars.expand(B, G)]
            g = out["g_posterior_samples"]
            beside = self.mode == "training" and self.overlap_prior and g.is_cuda and torch.is_grad_enabled() and \
                p_input.shape[0] * p_input.shape[2] <= self.overlap_prior_max_points and not torch.cuda.is_current_stream_capturing()
            buf_p = None
            if beside:
                buf_g, buf_p = self._prior_beside(g, lambda: self.pc_decoder(p_input, g, mode="inverse"))     # :138 beside :168
            else:
                buf_g = self.g_prior(g, mode="inverse")                                        # :138 / :182
            out["g_prior_samples"] = buf_g[0] + [g]
            out["g_prior_mus"] += buf_g[1]
            out["g_prior_logvars"] += buf_g[2]
            if self.mode == "training":
                out["p_prior_mus"], out["p_prior_logvars"] = self._base(g, p_input.shape[0], p_input.shape[2])
                if buf_p is None:
                    buf_p = self.pc_decoder(p_input, g, mode="inverse")                        # :168
                out["p_prior_samples"] = buf_p[0] + [p_input]
            else:
                out["p_prior_mus"], out["p_prior_logvars"] = self._base(g, p_input.shape[0], S)
                out["p_prior_samples"] = [self.reparameterize(out["p_prior_mus"][0], out["p_prior_logvars"][0])]   # :211
                buf_p = self.pc_decoder(out["p_prior_samples"][0], g, mode="direct")           # :212
                out["p_prior_samples"] += buf_p[0]
            out["p_prior_mus"] += buf_p[1]
            out["p_prior_logvars"] += buf_p[2]
        elif self.mode == "generating":                                                        # :218-256
            out["g_prior_mus"] = [self.g0_prior_mus.expand(B, G)]
            out["g_prior_logvars"] = [self.g0_prior_logvars.expand(B, G)]
            out["g_prior_samples"] = [self.reparameterize(out["g_prior_mus"][0], out["g_prior_logvars"][0])]
            buf_g = self.g_prior(out["g_prior_samples"][0], mode="direct")
            out["g_prior_samples"] += buf_g[0]
            out["g_prior_mus"] += buf_g[1]
            out["g_prior_logvars"] += buf_g[2]
            g = out["g_prior_samples"][-1]
            out["p_prior_mus"], out["p_prior_logvars"] = self._base(g, p_input.shape[0], S)
            out["p_prior_samples"] = [self.reparameterize(out["p_prior_mus"][0], out["p_prior_logvars"][0])]
            buf_p = self.pc_decoder(out["p_prior_samples"][0], g, mode="direct")
            out["p_prior_samples"] += buf_p[0]
            out["p_prior_mus"] += buf_p[1]
            out["p_prior_logvars"] += buf_p[2]
        return out

    def flatten_parameters(self):
        """Put the point decoder's and the prior flow's parameters into their flat stores (call after .cuda()); returns them."""
        return [self.pc_decoder.flatten_parameters(), self.g_prior.flatten_parameters()]
