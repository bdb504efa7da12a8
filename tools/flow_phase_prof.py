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
    for prec in (os.environ.get("DPF_PRECISION", "bf16x3"),):
        args.precision = prec
        L = args.layers
        dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, args.batch or 32)
        FW = int(os.environ.get('DPF_FLOW_WAVES', '8'))
        prof = torch.zeros((2 * FW, L, 16), dtype=torch.int64, device=dev)
        step = bench.make_step(dec, z, g, tgt_pm, L)
        for _ in range(3):
            step()
        handle.dpf_debug_set_prof(prof.data_ptr())
        step()
        torch.cuda.synchronize()
        handle.dpf_debug_set_prof(None)
        t = prof.cpu().numpy().astype(np.int64)
        d = np.diff(t[:, :, :8], axis=2)            # phases 0..6
        names = ["input fragment + G0 (input MFMAs, split A k0)", "G1-G4 chain A (24 MFMA)", "G5-G8 chain B (24 MFMA)",
                 "tail: contraction B", "half sums + transform + list stores", "wait for own DMA pieces (vmcnt 0)", "workgroup barrier"]
        print("== %s: cycles per layer (median over waves/layers) | per-wave layer period" % prec)
        for i, nme in enumerate(names):
            print("   %-38s median %7.0f  p90 %7.0f" % (nme, np.median(d[:, 1:, i]), np.percentile(d[:, 1:, i], 90)))
        period = np.diff(t[:, :, 0], axis=1)
        print("   layer period   median %7.0f  (total/layer incl. staging issue + B0 build)" % np.median(period))
        print("   wave0 first 3 layers raw deltas:", d[0, :3].tolist())
        # per-group stamps of the pipelined body: [1]=G0 end, 8,9,10 = end of G1,G2,G3, [2] = G4 end, 11,12,13 = end of G5,G6,G7, [3] = G8 end
        order = [1, 8, 9, 10, 2, 11, 12, 13, 3]
        gd = np.diff(t[:, :, order], axis=2)
        print("   cycles per group G1..G8 (median over waves/layers):", np.median(gd[:, 1:, :], axis=(0, 1)).tolist())
        print("   wave0 layer 2 groups:", gd[0, 2].tolist())
        # leaders (waves 0-3) and laggards (4-7) separately, with the gap from the end of a layer to the start of the next
        gap = t[:, 1:, 0] - t[:, :-1, 7]
        for nm, ws in (("leaders ", [0, 1, 2, 3]), ("laggards", [4, 5, 6, 7])):
            if max(ws) >= t.shape[0]:
                continue
            print("   %s: phases %s | gap to next layer %5.0f | period %5.0f" % (
                nm, [int(np.median(d[ws, 2:-1, i])) for i in range(7)], np.median(gap[ws, 1:]), np.median(period[ws, 1:])))
        odd = d[:, 1::2, :]                          # layers that end a buffer group (LPB = 2)
        print("   per wave (WG0 w0-7, WG1 w0-7), layers ending a group: own-DMA wait | barrier wait | start skew vs wave 0")
        for w in range(t.shape[0]):
            print("     wave %2d: %6.0f | %6.0f | %6.0f" % (w, np.median(odd[w, :, 5]), np.median(odd[w, :, 6]),
                                                   np.median(t[w, 1:, 0] - t[8 * (w // 8), 1:, 0])))


if __name__ == "__main__":
    main()
