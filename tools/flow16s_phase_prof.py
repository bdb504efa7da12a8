"""Per-phase s_memtime profile of the branch-split 16-point-tile kernel (flow16s_kernel; needs libdpf_hip_prof.so:
`make -C dpf_nets_amd/csrc prof`).   python tools/flow16s_phase_prof.py --batch 4"""
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
    L = args.layers
    B = args.batch or 4
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, B)
    CW = 4
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
    d = np.diff(t[:, :, :5], axis=2)
    names = ["meta + input fragment", "branch (4 + 24 MFMAs, splits, contraction, swaps)", "exchange write + workgroup barrier",
             "head fetch + transform + list stores"]
    print("== flow16s, B=%d L=%d: ticks per layer (100 MHz s_memtime x ~24 = cycles), median over waves and layers 2.." % (B, L))
    for i, nme in enumerate(names):
        print("   %-52s median %7.0f  p90 %7.0f" % (nme, np.median(d[:, 2:, i]), np.percentile(d[:, 2:, i], 90)))
    period = np.diff(t[:, :, 0], axis=1)
    print("   layer period median %7.0f" % np.median(period[:, 1:]))
    print("   per wave, layer 5:", d[:, 5].tolist())


if __name__ == "__main__":
    main()
