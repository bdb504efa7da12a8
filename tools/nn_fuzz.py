"""Randomised cross-check of the three Chamfer search kernels (matrix-core filter, SGPR-fed scan, LDS-staged scan) against each
other on the GPU: all are exact, so every output must agree bit for bit.  Shapes, scales, offsets, duplicates, lattices and
ragged tile counts are drawn at random; the CPU oracle pins a sample.    python tools/nn_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd._lib import lib, current_stream            # noqa: E402
from oracle import structural as S                           # noqa: E402


def run(L, fn, a, b, small_mode):
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    d1 = torch.empty((B, N), device="cuda"); d2 = torch.empty((B, M), device="cuda")
    i1 = torch.empty((B, N), dtype=torch.int32, device="cuda"); i2 = torch.empty((B, M), dtype=torch.int32, device="cuda")
    old = L.dpf_nn_small_mode(small_mode)
    try:
        if fn == "mfma":
            rc = L.dpf_nndistance_mfma(B, N, a.data_ptr(), M, b.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), None, 0, current_stream())
        else:
            rc = L.dpf_nndistance(B, N, a.data_ptr(), M, b.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), current_stream())
    finally:
        L.dpf_nn_small_mode(old)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return d1, i1, d2, i2


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = lib()
    t0, cases, pinned = time.time(), 0, 0
    while time.time() - t0 < budget:
        B = int(rng.integers(1, 9))
        N = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 257, 500, 1000, 1024, 2047, 2048, 2049, 2080, 2500, 3000, 4100]))
        M = int(rng.choice([1, 5, 32, 63, 65, 300, 777, 1024, 2048, 2111, 2500, 4096, 5000]))
        kind = rng.choice(["uniform", "normal", "lattice", "dups", "offset", "line", "tiny", "huge"])
        a = rng.random((B, N, 3), dtype=np.float32) - 0.5
        b = rng.random((B, M, 3), dtype=np.float32) - 0.5
        if kind == "normal":
            a = rng.standard_normal((B, N, 3)).astype(np.float32) * 0.2; b = rng.standard_normal((B, M, 3)).astype(np.float32) * 0.2
        elif kind == "lattice":
            a = np.round(a * 8) / 8; b = np.round(b * 8) / 8
        elif kind == "dups":
            b[:, M // 2:] = b[:, :M - M // 2]
            a[:, ::3] = b[:, :1]
        elif kind == "offset":
            a += np.float32(11.0); b += np.float32(11.0)
        elif kind == "line":
            a[..., 1:] = 0; b[..., 1:] = 0
        elif kind == "tiny":
            a *= np.float32(1e-6); b *= np.float32(1e-6)
        elif kind == "huge":
            a *= np.float32(1e6); b *= np.float32(1e6)
        ta, tb = torch.from_numpy(np.ascontiguousarray(a)).cuda(), torch.from_numpy(np.ascontiguousarray(b)).cuda()
        ref = run(L, "scan", ta, tb, 0)
        outs = {"mfma": run(L, "mfma", ta, tb, 0)}
        if max(N, M) <= 8192:
            outs["lds"] = run(L, "scan", ta, tb, 1)
        for name, o in outs.items():
            for x, y, w in zip(o, ref, ("d1", "i1", "d2", "i2")):
                if not torch.equal(x, y):
                    bad = (x != y).nonzero()[:3].tolist()
                    raise SystemExit("MISMATCH %s vs scan: %s B=%d N=%d M=%d kind=%s seed=%d case=%d at %s" % (name, w, B, N, M, kind, seed, cases, bad))
        if cases % 10 == 0 and B * N * M <= 4e7:
            o = S.nndistance(a, b)
            for x, y in zip(ref, o):
                assert np.array_equal(x.cpu().numpy().view(np.uint32) if x.dtype == torch.float32 else x.cpu().numpy(),
                                      y.view(np.uint32) if y.dtype == np.float32 else y), ("oracle", B, N, M, kind)
            pinned += 1
        cases += 1
    print("nn_fuzz: %d cases agree bit for bit across kernels (%d of them also pinned on the CPU oracle), seed %d" % (cases, pinned, seed))


if __name__ == "__main__":
    main()
