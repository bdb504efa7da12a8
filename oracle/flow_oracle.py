"""CPU restatement of dpf-nets' per-point conditional affine-coupling flow.

TEST INFRASTRUCTURE -- the checker, never the thing measured or shipped.  Only
tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it;
the product path (dpf_nets_amd) never does and fails loudly without its HIP
library.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against golden vectors captured from the reference's own PyTorch modules
imported on CPU (oracle/gen_golden.py -> tests/golden/*.npz).

The arithmetic is written out op by op in fp32 torch-CPU tensor ops (no
nn.Module, no F.batch_norm), in the order the reference evaluates it:

  SharedDot.forward              lib/networks/layers.py:40-45
  CondRealNVPFlow3D.forward      lib/networks/flows.py:95-117
  CondRealNVPFlow3DTriple        lib/networks/flows.py:151-160
  LocalCondRNVPDecoder.forward   lib/networks/decoders.py:54-72
  PointFlowNLL.forward           lib/networks/losses.py:11-15
  BatchNorm1d semantics          torch.nn.BatchNorm1d as used at flows.py:27,30,35,42
"""
import math

import numpy as np
import torch

from . import detrng

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FLOW_EPS = 1e-6

BRANCHES = ("mu", "logvar")


# --------------------------------------------------------------------------
# parameter naming (mirrors flows.py:25-93 so reference state dicts load)
# --------------------------------------------------------------------------
def layer_param_spec(F, G, warp_inds):
    """Ordered (key, shape, role) list of one CondRealNVPFlow3D's state dict."""
    nk = 3 - len(warp_inds)
    nw = len(warp_inds)
    spec = [("eps", (1,), "eps")]
    for br in BRANCHES:
        t0 = "T_%s_0" % br
        spec += [
            ("%s.%s_sd0.weight" % (t0, br), (1, F, nk), "sd0_w"),
            ("%s.%s_sd0_bn.weight" % (t0, br), (F,), "bn_gamma"),
            ("%s.%s_sd0_bn.bias" % (t0, br), (F,), "bn_beta"),
            ("%s.%s_sd0_bn.running_mean" % (t0, br), (F,), "bn_rm"),
            ("%s.%s_sd0_bn.running_var" % (t0, br), (F,), "bn_rv"),
            ("%s.%s_sd0_bn.num_batches_tracked" % (t0, br), (), "nbt"),
            ("%s.%s_sd1.weight" % (t0, br), (1, F, F), "sd1_w"),
            ("%s.%s_sd1_bn.running_mean" % (t0, br), (F,), "bn_rm"),
            ("%s.%s_sd1_bn.running_var" % (t0, br), (F,), "bn_rv"),
            ("%s.%s_sd1_bn.num_batches_tracked" % (t0, br), (), "nbt"),
        ]
        for s in ("w", "b"):
            tc = "T_%s_0_cond_%s" % (br, s)
            spec += [
                ("%s.%s_sd1_film_%s0.weight" % (tc, br, s), (F, G), "film0_w"),
                ("%s.%s_sd1_film_%s0_bn.weight" % (tc, br, s), (F,), "bn_gamma"),
                ("%s.%s_sd1_film_%s0_bn.bias" % (tc, br, s), (F,), "bn_beta"),
                ("%s.%s_sd1_film_%s0_bn.running_mean" % (tc, br, s), (F,), "bn_rm"),
                ("%s.%s_sd1_film_%s0_bn.running_var" % (tc, br, s), (F,), "bn_rv"),
                ("%s.%s_sd1_film_%s0_bn.num_batches_tracked" % (tc, br, s), (), "nbt"),
                ("%s.%s_sd1_film_%s1.weight" % (tc, br, s), (F, F), "film1_w"),
                ("%s.%s_sd1_film_%s1.bias" % (tc, br, s), (F,), "film1_b"),
            ]
        spec += [
            ("T_%s_1.%s_sd2.weight" % (br, br), (1, nw, F), "sd2_w"),
            ("T_%s_1.%s_sd2.bias" % (br, br), (1, nw), "sd2_b"),
        ]
    return spec


def make_layer_state(seed, F, G, warp_inds, final_std=0.05, film_std=0.05):
    """Deterministic, non-trivial weights for one coupling layer (numpy dict).

    Shapes/names follow flows.py:25-93; magnitudes follow the reference init
    (kaiming-uniform SharedDot layers.py:33, N(0, std) final layers
    flows.py:52-58) but BN affine/running stats are randomised so eval-mode BN
    is not the identity, and the final layers use a larger std than the
    reference's 0.01 so that mu/logvar are far from zero in the fixtures.
    """
    st = {}
    for k, shape, role in layer_param_spec(F, G, warp_inds):
        s = detrng.key(seed, k)
        n = int(np.prod(shape)) if len(shape) else 1
        if role == "eps":
            v = np.array([FLOW_EPS], dtype=np.float32)
        elif role == "sd0_w":
            b = math.sqrt(6.0 / (shape[1] * shape[2]))
            v = detrng.uniform_f32(s, shape, -b, b)
        elif role == "sd1_w":
            b = math.sqrt(6.0 / (shape[1] * shape[2]))
            v = detrng.uniform_f32(s, shape, -b, b) * 4.0
        elif role == "film0_w":
            b = 1.0 / math.sqrt(shape[1])
            v = detrng.uniform_f32(s, shape, -b, b)
        elif role == "film1_w":
            v = detrng.normal_f32(s, shape, 0.0, film_std)
        elif role == "film1_b":
            v = detrng.normal_f32(s, shape, 0.0, 0.05)
        elif role == "sd2_w":
            v = detrng.normal_f32(s, shape, 0.0, final_std)
        elif role == "sd2_b":
            v = detrng.normal_f32(s, shape, 0.0, 0.02)
        elif role == "bn_gamma":
            v = detrng.uniform_f32(s, shape, 0.5, 1.5)
        elif role == "bn_beta":
            v = detrng.normal_f32(s, shape, 0.0, 0.1)
        elif role == "bn_rm":
            v = detrng.normal_f32(s, shape, 0.0, 0.1)
        elif role == "bn_rv":
            v = detrng.uniform_f32(s, shape, 0.5, 1.5)
        elif role == "nbt":
            v = np.array(0, dtype=np.int64)
        else:
            raise KeyError(role)
        st[k] = v
    return st


TRIPLE_WARPS = {0: ([0], [1], [2]), 1: ([0, 1], [0, 2], [1, 2])}  # flows.py:129-148


def decoder_layer_plan(n_flows):
    """[(state-dict prefix, warp_inds)] in DIRECT order (decoders.py:50-51,58-64)."""
    plan = []
    for i in range(n_flows):
        warps = TRIPLE_WARPS[i % 2]
        for j in range(3):
            plan.append(("flows.%d.nvp%d." % (i, j + 1), list(warps[j])))
    return plan


def make_decoder_state(seed, n_flows, F, G, **kw):
    st = {}
    for prefix, warp in decoder_layer_plan(n_flows):
        lst = make_layer_state(detrng.key(seed, prefix), F, G, warp, **kw)
        for k, v in lst.items():
            st[prefix + k] = v
    return st


def to_torch(state):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}


def sub_state(state, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in state.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------
# arithmetic
# --------------------------------------------------------------------------
def shared_dot(weight, x, bias=None):
    """layers.py:40-45: (1,out,in) @ (B,1,in,N) -> (B,out,N), + bias (1,out)."""
    out = torch.matmul(weight, x.unsqueeze(1))
    if bias is not None:
        out = out + bias.unsqueeze(0).unsqueeze(3)
    return out.squeeze(1)


def batch_norm(x, rm, rv, gamma, beta, training, stats_out=None, key=None):
    """torch.nn.BatchNorm1d on (B,C,N) or (B,C).

    eval: (x - rm) / sqrt(rv + eps) * gamma + beta.
    train: batch mean / biased var over every dim but C; running stats updated
    with momentum 0.1 and the UNBIASED variance (recorded into stats_out[key]).
    """
    dims = [0] if x.dim() == 2 else [0, 2]
    shape = [1, -1] if x.dim() == 2 else [1, -1, 1]
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var = ((x - mean.view(shape)) ** 2).mean(dim=dims)
        if stats_out is not None:
            unbiased = var * (float(n) / float(max(n - 1, 1)))
            stats_out[key + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            stats_out[key + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased.detach()
    else:
        mean, var = rm, rv
    y = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS)
    if gamma is not None:
        y = y * gamma.view(shape) + beta.view(shape)
    return y


def _film(st, br, s, g, training, stats_out):
    """flows.py:33-45 / 68-80: Linear(G,F,no bias) . BN(batch dim) . Swish . Linear(F,F)."""
    tc = "T_%s_0_cond_%s" % (br, s)
    n0 = "%s.%s_sd1_film_%s0" % (tc, br, s)
    n1 = "%s.%s_sd1_film_%s1" % (tc, br, s)
    u = g @ st[n0 + ".weight"].t()
    u = batch_norm(u, st[n0 + "_bn.running_mean"], st[n0 + "_bn.running_var"],
                   st[n0 + "_bn.weight"], st[n0 + "_bn.bias"], training, stats_out, n0 + "_bn")
    u = u * torch.sigmoid(u)                                           # layers.py:9-10
    return u @ st[n1 + ".weight"].t() + st[n1 + ".bias"]


def _branch(st, br, x, g, training, stats_out):
    t0 = "T_%s_0" % br
    h = shared_dot(st["%s.%s_sd0.weight" % (t0, br)], x)               # flows.py:26/61
    h = batch_norm(h, st["%s.%s_sd0_bn.running_mean" % (t0, br)], st["%s.%s_sd0_bn.running_var" % (t0, br)],
                   st["%s.%s_sd0_bn.weight" % (t0, br)], st["%s.%s_sd0_bn.bias" % (t0, br)],
                   training, stats_out, "%s.%s_sd0_bn" % (t0, br))    # :27/62
    h = torch.relu(h)                                                  # :28/63
    h = shared_dot(st["%s.%s_sd1.weight" % (t0, br)], h)               # :29/64
    h = batch_norm(h, st["%s.%s_sd1_bn.running_mean" % (t0, br)], st["%s.%s_sd1_bn.running_var" % (t0, br)],
                   None, None, training, stats_out, "%s.%s_sd1_bn" % (t0, br))  # :30/65 affine=False
    cw = _film(st, br, "w", g, training, stats_out)
    cb = _film(st, br, "b", g, training, stats_out)
    eps = st["eps"]
    h = (eps + torch.exp(cw.unsqueeze(2))) * h + cb.unsqueeze(2)       # :100-101 / :105-106
    h = torch.relu(h)                                                  # :48/83
    return shared_dot(st["T_%s_1.%s_sd2.weight" % (br, br)], h,
                      st["T_%s_1.%s_sd2.bias" % (br, br)])             # :49/84


def coupling_layer(st, p, g, mode, warp_inds, training=False, stats_out=None):
    """CondRealNVPFlow3D.forward (flows.py:95-117).  p (B,3,N), g (B,G)."""
    keep = [c for c in (0, 1, 2) if c not in warp_inds]
    x = p[:, keep, :].contiguous()
    logvar = torch.zeros_like(p)
    mu = torch.zeros_like(p)
    o_lv = _branch(st, "logvar", x, g, training, stats_out)
    logvar[:, warp_inds, :] = o_lv / (1.0 + o_lv.abs())                # softsign, :99
    mu[:, warp_inds, :] = _branch(st, "mu", x, g, training, stats_out)  # :104
    scale = torch.sqrt(st["eps"] + torch.exp(logvar))
    if mode == "direct":
        p_out = scale * p + mu                                         # :113
    elif mode == "inverse":
        p_out = (p - mu) / scale                                       # :115
    else:
        raise ValueError(mode)
    return p_out, mu, logvar


def decoder(state, n_flows, p, g, mode, training=False, stats_out=None, n_layers=None):
    """LocalCondRNVPDecoder.forward (decoders.py:54-72) flattened over the
    triples (flows.py:151-160).  Lists come back in DIRECT order for both modes.
    n_layers < 3*n_flows runs only the first n_layers direct-order layers (the
    BASELINE metric's L=14 is the first 14 layers of n_flows=5)."""
    plan = decoder_layer_plan(n_flows)
    if n_layers is not None:
        plan = plan[:n_layers]
    L = len(plan)
    ps, mus, lvs = [None] * L, [None] * L, [None] * L
    order = range(L) if mode == "direct" else range(L - 1, -1, -1)
    cur = p
    for li in order:
        prefix, warp = plan[li]
        so = {} if stats_out is not None else None
        cur, m, lv = coupling_layer(sub_state(state, prefix), cur, g, mode, warp, training, so)
        if so is not None:
            for k, v in so.items():
                stats_out[prefix + k] = v
        ps[li], mus[li], lvs[li] = cur, m, lv
    return ps, mus, lvs


def point_flow_nll(samples, mus, logvars):
    """PointFlowNLL.forward (losses.py:11-15)."""
    s0 = samples[0]
    tot = sum(logvars) + (s0 - mus[0]) ** 2 / torch.exp(logvars[0])
    return 0.5 * (tot.sum() / s0.shape[0] + math.log(2.0 * math.pi) * s0.shape[1] * s0.shape[2])


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synthetic_inputs(seed, B, N, G):
    """targets U[-0.25,0.25]^3 (B,3,N); base samples N(0, e^-3.6); g ~ N(0,1)."""
    tgt = detrng.uniform_f32(detrng.key(seed, "target"), (B, 3, N), -0.25, 0.25)
    z = detrng.normal_f32(detrng.key(seed, "base"), (B, 3, N), 0.0, math.exp(-1.8))
    g = detrng.normal_f32(detrng.key(seed, "latent"), (B, G), 0.0, 1.0)
    return tgt, z, g
