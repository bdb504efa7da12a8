"""Per-phase s_memtime profile of the DUAL flow body (csrc/flow_dual.h; needs libdpf_hip_prof.so and DPF_FLOW_DUAL=1)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DPF_FLOW_DUAL"] = "1"
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
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, args.batch or 32)
    prof = torch.zeros((8, L, 16), dtype=torch.int64, device=dev)
    step = bench.make_step(dec, z, g, tgt_pm, L)
    for _ in range(3):
        step()
    handle.dpf_debug_set_prof(prof.data_ptr())
    step()
    torch.cuda.synchronize()
    handle.dpf_debug_set_prof(None)
    t = prof.cpu().numpy().astype(np.int64)
    d = np.diff(t[:, :, :6], axis=2)
    names = ["phase 1 (X chain A | Y tail + S0)", "barrier", "ds_write of layer n+2", "phase 2 (X chain B | Y chain A)",
             "phase 3 (X tail + S0 | Y chain B)"]
    for i, nme in enumerate(names):
        print("   %-40s median %6.0f  p90 %6.0f" % (nme, np.median(d[:, 1:-1, i]), np.percentile(d[:, 1:-1, i], 90)))
    print("   layer period median %6.0f" % np.median(np.diff(t[:, :, 0], axis=1)))
    print("   wave 0 layers 1..3:", d[0, 1:4].tolist())
    if t[0, 2, 6] > 0:       # -DDPF_PROFILE_FINE: a stamp after every group
        order = [0, 6, 7, 8, 1, 3, 9, 10, 11, 4, 12, 13, 14, 5]
        gd = np.diff(t[:, :, order], axis=2)
        lab = ["1a sa0|s9", "1b sa1|s10", "1c sa2|s11", "1d sa3|s0", "(barrier+write)", "2a", "2b", "2c", "2d", "3a s9|sb0", "3b s10|sb1", "3c s11|sb2", "3d s0|sb3"]
        med = np.median(gd[:, 2:-1, :], axis=(0, 1))
        for k, v in zip(lab, med):
            print("      %-18s %6.0f" % (k, v))


if __name__ == "__main__":
    main()
