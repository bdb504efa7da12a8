"""Per-phase s_memtime profile of tbwd2 (needs libdpf_hip_prof.so: make -C dpf_nets_amd/csrc prof)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", "libdpf_hip_prof.so")
from dpf_nets_amd import networks as nets, synthetic as SY
from dpf_nets_amd.networks import train_engine
B, N, G = 32, 2048, 128
h = _lib.lib()
h.dpf_debug_set_tprof.argtypes = [ctypes.c_void_p]
dec = nets.LocalCondRNVPDecoder(1, 64, G).cuda().train()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
# r05 stamps (first tile of every wave): TP(0) tile top .. TP(6) dW1 issued; [11] prologue done, [12] entry, [13] exit
names = ["tile top", "input fragment", "fwd recompute", "dh1 + fragments", "transposes, W1^T chain, mask, u_k", "swapped chain + sums", "dW1"]
for prec in ("f16x3",):
    train_engine.TRAIN_PRECISION = prec
    prof = torch.zeros((16, 14), dtype=torch.int64, device="cuda")
    for it in range(3):
        if it == 2: h.dpf_debug_set_tprof(prof.data_ptr())
        x = torch.from_numpy(tgt).cuda().requires_grad_(True)
        ps, mus, lvs = dec(x, torch.from_numpy(g).cuda(), mode="inverse")
        (ps[0].square().mean() + sum(lvs).mean()).backward()
    torch.cuda.synchronize()
    h.dpf_debug_set_tprof(None)
    t = prof.cpu().numpy()
    d = np.diff(t[:, :7], axis=1)
    print(prec, "kernel entry -> after the prologue:", np.median(t[:, 11] - t[:, 12]), " prologue end -> loop:", np.median(t[:, 0] - t[:, 11]),
          " first tile:", np.median(t[:, 6] - t[:, 0]), " first tile done -> exit (second tile + reduction):", np.median(t[:, 13] - t[:, 6]),
          " whole kernel (this wave):", np.median(t[:, 13] - t[:, 12]))
    print("   second tile (stamps 7, 8):", "median %.0f" % np.median(t[:, 8] - t[:, 7]), " per wave of workgroup 0:", " ".join("%5d" % v for v in (t[:8, 8] - t[:8, 7])),
          "\n   second tile done -> exit (reduction, incl. the wait for the slowest wave):", " ".join("%5d" % v for v in (t[:8, 13] - t[:8, 8])),
          "\n      of which: arrival at the reduction's first barrier", " ".join("%5d" % v for v in (t[:8, 9] - t[:8, 8])), "| two dW1 rounds", " ".join("%5d" % v for v in (t[:8, 10] - t[:8, 9])), "| sums round + exit", " ".join("%5d" % v for v in (t[:8, 13] - t[:8, 10])),
          "\n   entry -> exit per wave:", " ".join("%5d" % v for v in (t[:8, 13] - t[:8, 12])))
    for i in range(6):
        print("   %-16s median %8.0f  min %8.0f  max %8.0f   per wave of workgroup 0: %s" % (names[i + 1], np.median(d[:, i]), d[:, i].min(), d[:, i].max(), " ".join("%5d" % v for v in d[:8, i])))
    print("   arrival at the phase boundaries relative to the workgroup's first wave (workgroup 0):")
    for i in range(7):
        print("      TP(%2d) %s" % (i, " ".join("%6d" % (v - t[:8, i].min()) for v in t[:8, i])))
