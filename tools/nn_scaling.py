"""Chamfer nn kernel: time vs problem size, to separate the per-pair cost from fixed overheads."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import synthetic as SY                         # noqa: E402
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK   # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    for (B, N, M) in ((32, 2048, 2048), (64, 2048, 2048), (32, 4096, 4096), (32, 2048, 4096), (128, 2048, 2048), (32, 8192, 8192)):
        a = torch.from_numpy(SY.uniform_f32(1, (B, N, 3), -0.25, 0.25)).cuda()
        b = torch.from_numpy(SY.uniform_f32(2, (B, M, 3), -0.25, 0.25)).cuda()
        t = timed(lambda: BK.NNDistance(a, b))
        pairs = 2.0 * B * N * M
        print("B=%d N=%d M=%d: %.1f us, %.3e pairs/s, %.2f ns per pair per SIMD" % (B, N, M, t, pairs / t * 1e6, t * 1e3 * 1024 * 64 / pairs), flush=True)


if __name__ == "__main__":
    main()
