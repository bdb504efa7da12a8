"""Deterministic synthetic weights and inputs for benchmarks, smoke runs and examples.

Version-stable counter-based streams (splitmix64 -> uniform -> Box-Muller), reference-shaped
coupling-layer state dicts (names/shapes of lib/networks/flows.py:25-93) and the synthetic
clouds of SURVEY.md section 8(d).  tests/test_synthetic.py checks that this module and the test
oracle's own generator (oracle/detrng.py, oracle/flow_oracle.py) emit identical values, so
fixtures and benchmarks are keyed on the same streams without the product importing oracle/.
"""
import math
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def key(seed, name=""):
    """Fold a string stream name into an integer seed."""
    return (int(seed) * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF


def uniform(seed, n, lo=0.0, hi=1.0):
    """n float64 uniforms in [lo, hi)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(1))
        bits = _splitmix64(idx ^ base)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def normal(seed, n, mean=0.0, std=1.0):
    """n float64 normals via Box-Muller on two uniform streams."""
    u1 = uniform(key(seed, "bm1"), n)
    u2 = uniform(key(seed, "bm2"), n)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return mean + std * r * np.cos(2.0 * np.pi * u2)


def uniform_f32(seed, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    return uniform(seed, n, lo, hi).astype(np.float32).reshape(shape)


def normal_f32(seed, shape, mean=0.0, std=1.0):
    n = int(np.prod(shape))
    return normal(seed, n, mean, std).astype(np.float32).reshape(shape)


BRANCHES = ("mu", "logvar")
FLOW_EPS = 1e-6


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
        s = key(seed, k)
        n = int(np.prod(shape)) if len(shape) else 1
        if role == "eps":
            v = np.array([FLOW_EPS], dtype=np.float32)
        elif role == "sd0_w":
            b = math.sqrt(6.0 / (shape[1] * shape[2]))
            v = uniform_f32(s, shape, -b, b)
        elif role == "sd1_w":
            b = math.sqrt(6.0 / (shape[1] * shape[2]))
            v = uniform_f32(s, shape, -b, b) * 4.0
        elif role == "film0_w":
            b = 1.0 / math.sqrt(shape[1])
            v = uniform_f32(s, shape, -b, b)
        elif role == "film1_w":
            v = normal_f32(s, shape, 0.0, film_std)
        elif role == "film1_b":
            v = normal_f32(s, shape, 0.0, 0.05)
        elif role == "sd2_w":
            v = normal_f32(s, shape, 0.0, final_std)
        elif role == "sd2_b":
            v = normal_f32(s, shape, 0.0, 0.02)
        elif role == "bn_gamma":
            v = uniform_f32(s, shape, 0.5, 1.5)
        elif role == "bn_beta":
            v = normal_f32(s, shape, 0.0, 0.1)
        elif role == "bn_rm":
            v = normal_f32(s, shape, 0.0, 0.1)
        elif role == "bn_rv":
            v = uniform_f32(s, shape, 0.5, 1.5)
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
        lst = make_layer_state(key(seed, prefix), F, G, warp, **kw)
        for k, v in lst.items():
            st[prefix + k] = v
    return st




def synthetic_inputs(seed, B, N, G):
    """targets U[-0.25,0.25]^3 (B,3,N); base samples N(0, e^-3.6); g ~ N(0,1)."""
    tgt = uniform_f32(key(seed, "target"), (B, 3, N), -0.25, 0.25)
    z = normal_f32(key(seed, "base"), (B, 3, N), 0.0, math.exp(-1.8))
    g = normal_f32(key(seed, "latent"), (B, G), 0.0, 1.0)
    return tgt, z, g
