"""Per-phase s_memtime profile of the fused flow kernel (needs libdpf_hip_prof.so, a -DDPF_PROFILE
build of csrc/flow.hip).  Prints the median cycles each wave spends per phase per layer."""
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
    extra = sys.argv[1:]
    sys.argv = ["bench.py"] + extra          # e.g. --batch 16: one wave per SIMD
    args = bench.parse()
    dev = torch.device("cuda", 0)
    handle = _lib.lib()
    handle.dpf_debug_set_prof.argtypes = [ctypes.c_void_p]
    FW = 8
    for prec in ("bf16x3",):
        args.precision = prec
        L = args.layers
        dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
        prof = torch.zeros((2 * FW, L, 8), dtype=torch.int64, device=dev)
        step = bench.make_step(dec, z, g, tgt_pm, L)
        for _ in range(3):
            step()
        handle.dpf_debug_set_prof(prof.data_ptr())
        step()
        torch.cuda.synchronize()
        handle.dpf_debug_set_prof(None)
        t = prof.cpu().numpy().astype(np.int64)
        d = np.diff(t[:, :, :7], axis=2)            # phases 0..5
        # stamps 1 and 2 are taken inside branch_tile, which runs twice per layer: the second call (branch mu) overwrites them
        names = ["branch 0 (all) + branch 1 input/split", "branch 1 chain", "branch 1 epilogue", "transform + list stores",
                 "end-of-layer DMA wait + barrier", "-"]
        print("== %s: cycles per layer (median over waves/layers) | per-wave layer period" % prec)
        for i, nme in enumerate(names):
            print("   %-38s median %7.0f  p90 %7.0f" % (nme, np.median(d[:, 1:, i]), np.percentile(d[:, 1:, i], 90)))
        period = np.diff(t[:, :, 0], axis=1)
        print("   layer period   median %7.0f  (total/layer incl. staging issue + B0 build)" % np.median(period))
        print("   wave0 first 3 layers raw deltas:", d[0, :3].tolist())


if __name__ == "__main__":
    main()
