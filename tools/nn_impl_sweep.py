"""Brute-force vs matrix-core-filtered Chamfer nn over problem sizes (to place the default crossover)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import synthetic as SY                         # noqa: E402
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK   # noqa: E402
from tools.nn_scaling import timed                               # noqa: E402


def main():
    shapes = [(32, 64, 64), (32, 128, 128), (32, 256, 256), (32, 512, 512), (32, 1024, 1024), (32, 2048, 2048), (4, 2048, 2048),
              (1, 2048, 2048), (8, 8192, 8192), (2, 16384, 16384), (32, 2048, 100), (32, 100, 2048), (32, 2500, 2500)]
    for (B, N, M) in shapes:
        a = torch.from_numpy(SY.uniform_f32(1, (B, N, 3), -0.25, 0.25)).cuda()
        b = torch.from_numpy(SY.uniform_f32(2, (B, M, 3), -0.25, 0.25)).cuda()
        res = {}
        for impl in ("brute", "mfma"):
            BK.NN_IMPL = impl
            res[impl] = timed(lambda: BK.NNDistance(a, b))
        print("B=%d N=%d M=%d: brute %.1f us  mfma %.1f us  (%.2fx)" % (B, N, M, res["brute"], res["mfma"], res["brute"] / res["mfma"]), flush=True)


def distributions():
    """cfg-2 size on data the filter likes less: Gaussian blobs, a thin surface, heavy exact ties (coarse lattice),
    identical clouds"""
    import numpy as np
    B, N = 32, 2048
    rng = np.random.default_rng(0)
    sets = {}
    g1 = (rng.standard_normal((B, N, 3)) * 0.1).astype(np.float32); g2 = (rng.standard_normal((B, N, 3)) * 0.1).astype(np.float32)
    sets["gauss"] = (g1, g2)
    u = rng.uniform(-0.25, 0.25, (2, B, N, 3)).astype(np.float32); u[..., 2] *= 1e-3
    sets["thin slab"] = (u[0], u[1])
    l = np.round(rng.uniform(-0.25, 0.25, (2, B, N, 3)) * 16).astype(np.float32) / 16
    sets["lattice 1/16 (ties)"] = (l[0], l[1])
    sets["identical clouds"] = (g1, g1.copy())
    far = g2 + np.float32(50.0)
    sets["far offset (+50)"] = (g1 + np.float32(50.0), far)
    for name, (a, b) in sets.items():
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        res, outs = {}, {}
        for impl in ("brute", "mfma"):
            BK.NN_IMPL = impl
            outs[impl] = BK.NNDistance(ta, tb)
            res[impl] = timed(lambda: BK.NNDistance(ta, tb))
        same = all(torch.equal(x, y) for x, y in zip(outs["brute"], outs["mfma"]))
        print("%-22s brute %.1f us  mfma %.1f us  (%.2fx)  identical results: %s" % (name, res["brute"], res["mfma"], res["brute"] / res["mfma"], same), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        distributions()
        sys.exit(0)
    main()
