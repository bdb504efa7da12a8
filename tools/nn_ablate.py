"""Timing experiments on the Chamfer VALU scan at a rank-sized batch: the kernel with one piece compiled out
(`make -C dpf_nets_amd/csrc nn_ablate ABLATE=<mask>` -> libdpf_nn<mask>.so; results are garbage, durations only).
    python tools/nn_ablate.py <mask> [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib  # noqa: E402

mask = int(sys.argv[1])
if mask:
    _lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", "libdpf_nn%d.so" % mask)
import bench                                                     # noqa: E402
from dpf_nets_amd import synthetic as SY                         # noqa: E402
from dpf_nets_amd._lib import lib, current_stream                # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = 2048
L = lib()
a = torch.from_numpy(SY.uniform_f32(1, (B, N, 3), -0.25, 0.25)).cuda()
b = torch.from_numpy(SY.uniform_f32(2, (B, N, 3), -0.25, 0.25)).cuda()
d1 = torch.empty((B, N), dtype=torch.float32, device="cuda"); d2 = torch.empty_like(d1)
i1 = torch.empty((B, N), dtype=torch.int32, device="cuda"); i2 = torch.empty_like(i1)
args = (B, N, a.data_ptr(), N, b.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr())
fn = lambda: L.dpf_nndistance(*args, current_stream())   # noqa: E731
fn()
torch.cuda.synchronize()
print("ablate mask %2d  KS=%s  B=%d: %.2f us" % (mask, os.environ.get("DPF_NN_KS", "auto"), B, bench.time_kernel(fn)), flush=True)
