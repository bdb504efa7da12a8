"""pairwise_CD (lib/networks/utils.py:90-117) at evaluation size: the one-launch matrix kernel vs the per-row loop
(one zero-stride Chamfer launch + one reduction per row -- what r01 shipped).  Usage: pairwise_bench.py [N1 N2 n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd._lib import lib, check, current_stream      # noqa: E402
from dpf_nets_amd.networks.utils import pairwise_CD           # noqa: E402

N1, N2, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 2048)
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(N1, n, 3, device="cuda", generator=g) * 0.2
b = torch.randn(N2, n, 3, device="cuda", generator=g) * 0.2


def row_loop():
    cds = torch.empty((N1, N2), device="cuda")
    d1 = torch.empty((N2, n), device="cuda"); d2 = torch.empty((N2, n), device="cuda")
    i1 = torch.empty((N2, n), dtype=torch.int32, device="cuda"); i2 = torch.empty((N2, n), dtype=torch.int32, device="cuda")
    st = current_stream()
    for i in range(N1):
        check(lib().dpf_nndistance_strided_auto(N2, n, a[i].data_ptr(), 0, n, b.data_ptr(), n * 3, d1.data_ptr(), i1.data_ptr(),
                                                d2.data_ptr(), i2.data_ptr(), st), "strided")
        check(lib().dpf_chamfer_reduce(N2, n, n, d1.data_ptr(), d2.data_ptr(), cds[i].data_ptr(), st), "reduce")
    return cds


for name, fn in (("one launch", lambda: pairwise_CD(a, b)), ("row loop  ", row_loop)):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%s N1=%d N2=%d n=%d: %.2f ms  %.3g pair evaluations/s" % (name, N1, N2, n, dt * 1e3, 2.0 * N1 * N2 * n * n / dt))
    if name.startswith("one"):
        ref = out
    else:
        print("max rel diff vs one launch: %.2e" % float(((out - ref).abs() / ref.abs()).max()))
