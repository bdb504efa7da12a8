"""Chamfer nn at the batch sizes a rank of an 8-GPU job holds (B = 2..16 clouds of 2048 points): VALU scan vs the matrix-core
filter with small workgroups (DPF_NNM_QW = 4 | 8 | 16 picks the filter kernel's workgroup size, one process per setting).
Kernel durations from HIP events around a captured graph of back-to-back launches (bench.time_kernel): no host time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib as _L                             # noqa: E402
if os.environ.get("DPF_LIB"):                                    # a timing-experiment build of the library
    _L.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", os.environ["DPF_LIB"])
import bench                                                     # noqa: E402
from dpf_nets_amd import synthetic as SY                         # noqa: E402
from dpf_nets_amd._lib import lib, current_stream                # noqa: E402


def main():
    L = lib()
    shapes = [(2, 2048, 2048), (4, 2048, 2048), (8, 2048, 2048), (16, 2048, 2048), (4, 2500, 2500), (2, 8192, 8192), (32, 2048, 2048)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
    for (B, N, M) in shapes:
        a = torch.from_numpy(SY.uniform_f32(1, (B, N, 3), -0.25, 0.25)).cuda()
        b = torch.from_numpy(SY.uniform_f32(2, (B, M, 3), -0.25, 0.25)).cuda()
        d1 = torch.empty((B, N), dtype=torch.float32, device="cuda"); d2 = torch.empty((B, M), dtype=torch.float32, device="cuda")
        i1 = torch.empty((B, N), dtype=torch.int32, device="cuda"); i2 = torch.empty((B, M), dtype=torch.int32, device="cuda")
        args = (B, N, a.data_ptr(), M, b.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr())
        res, outs = {}, {}
        def scan(mode):
            def f():
                L.dpf_nn_small_mode(mode)
                L.dpf_nndistance(*args, current_stream())
                L.dpf_nn_small_mode(-1)
            return f
        for name, fn in (("brute", scan(0)), ("lds", scan(1)),
                         ("mfma", lambda: L.dpf_nndistance_mfma(*args, None, 0, current_stream())),
                         ("auto", lambda: L.dpf_nndistance_auto(*args, current_stream()))):
            fn()
            torch.cuda.synchronize()
            outs[name] = (d1.clone(), i1.clone(), d2.clone(), i2.clone())
            res[name] = bench.time_kernel(fn)
        same = all(torch.equal(x, y) for x, y in zip(outs["brute"], outs["mfma"]))
        same = same and all(torch.equal(x, y) for x, y in zip(outs["brute"], outs["lds"]))
        print("QW=%s B=%d N=%d M=%d: brute %.1f us  lds %.1f us  mfma %.1f us  auto %.1f us  identical: %s"
              % (os.environ.get("DPF_NNM_QW", "auto"), B, N, M, res["brute"], res["lds"], res["mfma"], res["auto"], same), flush=True)


if __name__ == "__main__":
    main()
