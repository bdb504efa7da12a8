"""Per-phase s_memtime profile of the 16-point-tile flow kernel (needs libdpf_hip_prof.so: `make -C dpf_nets_amd/csrc prof`).
    python tools/flow16_phase_prof.py --batch 4        (DPF_FLOW16_CW=2|4 picks the workgroup shape)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib  # noqa: E402

_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", "libdpf_hip_prof.so")
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    handle = _lib.lib()
    handle.dpf_debug_set_prof.argtypes = [ctypes.c_void_p]
    handle.dpf_flow_set_tile16(1)
    L = args.layers
    B = args.batch or 4
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, B)
    CW = int(os.environ.get("DPF_FLOW16_CW", "4"))
    prof = torch.zeros((2 * CW, L, 8), dtype=torch.int64, device=dev)
    ks = bench.make_kernels(dec, z, g, tgt_pm, L, args.precision)
    for _ in range(3):
        for k in ks:
            k()
    handle.dpf_debug_set_prof(prof.data_ptr())
    ks[1]()
    torch.cuda.synchronize()
    handle.dpf_debug_set_prof(None)
    t = prof.cpu().numpy().astype(np.int64)
    d = np.diff(t[:, :, :7], axis=2)
    names = ["G0: input MFMAs + A's splits", "chain A (24 MFMA) + B's splits", "chain B (24 MFMA) + A's contraction",
             "workgroup barrier", "head fetch + B's contraction + quad sums", "transform + list stores"]
    print("== flow16, B=%d L=%d CW=%d: cycles per layer, median over waves and layers 2.." % (B, L, CW))
    for i, nme in enumerate(names):
        print("   %-42s median %7.0f  p90 %7.0f" % (nme, np.median(d[:, 2:, i]), np.percentile(d[:, 2:, i], 90)))
    period = np.diff(t[:, :, 0], axis=1)
    gap = t[:, 1:, 0] - t[:, :-1, 6]
    print("   layer period median %7.0f | end of a layer -> start of the next %5.0f (input fragment, meta)" % (np.median(period[:, 1:]), np.median(gap[:, 1:])))
    print("   wave 0, layers 2-4 raw:", d[0, 2:5].tolist())


if __name__ == "__main__":
    main()
