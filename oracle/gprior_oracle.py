"""CPU restatement of dpf-nets' latent prior flow: RealNVPFlow, RealNVPFlowCouple and GlobalRNVPDecoder.

TEST INFRASTRUCTURE -- the checker, never the thing measured or shipped.  Only tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke() may import it.

Parity status: PINNED.  tests/test_oracle_golden.py checks it against vectors captured from the reference's own
`GlobalRNVPDecoder` imported on CPU (oracle/gen_golden.py -> tests/golden/gprior.npz), eval and train mode.

  RealNVPFlow.__init__/forward         lib/networks/flows.py:163-213
      keep = every index not in warp_inds;   for T in (mu, logvar):
          T_0 = Linear(|keep| -> n_features, no bias) . BatchNorm1d . Swish . Linear(n_features -> |warp|, bias)
      logvar[:, warp] = log(eps + exp(T_logvar_0(g[:, keep]))),  mu[:, warp] = T_mu_0(g[:, keep]),  zero elsewhere
      direct: g_out = exp(0.5 logvar) g + mu          inverse: g_out = exp(-0.5 logvar) (g - mu)
  RealNVPFlowCouple                    lib/networks/flows.py:216-243
      pattern 0: nvp1 warps the even indices, nvp2 the odd ones; pattern 1: first half / second half;
      direct runs nvp1 then nvp2, inverse nvp2 then nvp1; returns [g1, g2], [mu1, mu2], [logvar1, logvar2]
  GlobalRNVPDecoder                    lib/networks/decoders.py:7-38
      couples with pattern i % 2; inverse walks them backwards and prepends: every list is in DIRECT order
"""
import math

import numpy as np
import torch

from . import detrng
from .flow_oracle import batch_norm

EPS = 1e-6          # RealNVPFlow's `eps` buffer (flows.py:164,171)


def warp_indices(G, pattern, k):
    """warp_inds of nvp{k+1} of a couple (flows.py:224-233)."""
    idx = np.arange(G)
    if pattern == 0:
        return list(idx[::2]) if k == 0 else list(idx[1::2])
    return list(idx[:G // 2]) if k == 0 else list(idx[G // 2:])


def step_plan(n_flows, G):
    """[(module prefix, warp_inds, keep_inds)] of the 2*n_flows coupling steps in DIRECT order."""
    plan = []
    for i in range(n_flows):
        for k in range(2):
            warp = warp_indices(G, i % 2, k)
            keep = [j for j in range(G) if j not in set(warp)]
            plan.append(("flows.%d.nvp%d." % (i, k + 1), warp, keep))
    return plan


def make_gprior_state(seed, n_flows, n_features, G, final_std=0.05):
    """Deterministic non-trivial weights (numpy dict, names as the reference's state_dict): nn.Linear's default
    magnitude for the first map, N(0, final_std) for the second (the reference's 0.01 would leave mu and logvar
    nearly zero), BatchNorm affine / running statistics randomised so eval BN is not the identity."""
    st = {}
    for prefix, warp, keep in step_plan(n_flows, G):
        st[prefix + "eps"] = np.array([EPS], dtype=np.float32)
        for br in ("mu", "logvar"):
            base = "%sT_%s_0.%s_" % (prefix, br, br)
            b = 1.0 / math.sqrt(len(keep))
            st[base + "mlp0.weight"] = detrng.uniform_f32(detrng.key(seed, base + "w0"), (n_features, len(keep)), -b, b)
            st[base + "mlp0_bn.weight"] = detrng.uniform_f32(detrng.key(seed, base + "g"), (n_features,), 0.5, 1.5)
            st[base + "mlp0_bn.bias"] = detrng.normal_f32(detrng.key(seed, base + "b"), (n_features,), 0.0, 0.1)
            st[base + "mlp0_bn.running_mean"] = detrng.normal_f32(detrng.key(seed, base + "rm"), (n_features,), 0.0, 0.1)
            st[base + "mlp0_bn.running_var"] = detrng.uniform_f32(detrng.key(seed, base + "rv"), (n_features,), 0.5, 1.5)
            st[base + "mlp0_bn.num_batches_tracked"] = np.array(0, dtype=np.int64)
            st[base + "mlp1.weight"] = detrng.normal_f32(detrng.key(seed, base + "w1"), (len(warp), n_features), 0.0, final_std)
            st[base + "mlp1.bias"] = detrng.normal_f32(detrng.key(seed, base + "b1"), (len(warp),), 0.0, final_std)
    return st


def gprior_inputs(seed, B, G):
    """(B, G) latent codes at unit scale (posterior samples / N(0,1) prior draws)."""
    return detrng.normal_f32(detrng.key(seed, "gprior_g"), (B, G), 0.0, 1.0)


def _net(st, base, x, training, stats_out):
    """Linear . BatchNorm1d . Swish . Linear (flows.py:176-181)."""
    h = x @ st[base + "mlp0.weight"].t()
    h = batch_norm(h, st[base + "mlp0_bn.running_mean"], st[base + "mlp0_bn.running_var"], st[base + "mlp0_bn.weight"],
                   st[base + "mlp0_bn.bias"], training, stats_out, base + "mlp0_bn")
    h = h * torch.sigmoid(h)
    return h @ st[base + "mlp1.weight"].t() + st[base + "mlp1.bias"]


def realnvp_flow(st, prefix, g, mode, warp, keep, training=False, stats_out=None):
    """flows.py:198-213.  st: dict of torch tensors."""
    logvar = torch.zeros_like(g)
    mu = torch.zeros_like(g)
    gk = g[:, keep].contiguous()
    logvar[:, warp] = torch.log(st[prefix + "eps"] + torch.exp(_net(st, prefix + "T_logvar_0.logvar_", gk, training, stats_out)))
    mu[:, warp] = _net(st, prefix + "T_mu_0.mu_", gk, training, stats_out)
    if mode == "direct":
        out = torch.exp(0.5 * logvar) * g + mu
    elif mode == "inverse":
        out = torch.exp(-0.5 * logvar) * (g - mu)
    else:
        raise ValueError(mode)
    return out, mu, logvar


def global_rnvp_decoder(st, n_flows, g, mode, training=False, stats_out=None):
    """decoders.py:21-38 with flows.py:235-243 inlined: three lists of 2*n_flows (B,G) tensors in DIRECT order."""
    plan = step_plan(n_flows, g.shape[1])
    S = len(plan)
    gs, mus, lvs = [None] * S, [None] * S, [None] * S
    order = range(S) if mode == "direct" else range(S - 1, -1, -1)
    cur = g
    for s in order:
        prefix, warp, keep = plan[s]
        cur, mus[s], lvs[s] = realnvp_flow(st, prefix, cur, mode, warp, keep, training, stats_out)
        gs[s] = cur
    return gs, mus, lvs
