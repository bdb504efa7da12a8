"""Version-stable deterministic random streams for fixtures and synthetic inputs.

TEST INFRASTRUCTURE. Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import anything under oracle/.

numpy's Generator makes no cross-version stream guarantee, and torch's RNG
differs between CPU and GPU, so golden fixtures are keyed on this
counter-based generator instead: splitmix64 over (seed, index) -> uniform
doubles -> Box-Muller normals.  Pure uint64/float64 numpy arithmetic, so the
stream is identical wherever it runs (this container, the GPU box).
"""
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
