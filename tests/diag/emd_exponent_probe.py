"""Diagnostic: the matrix-core passes' exp2 arguments (dpf_debug_emd_exponents) against float64, per level; with
DPF_EMD_DEBUG_SLOTS=<mask> only the named K slots contribute (bit k: slot k of MFMA 1, bit 16 + k: of MFMA 2) -- compared with the
CPU emulation of the same slots (tests/diag/emd_grid_emulation.py).   emd_exponent_probe.py [kind]"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dpf_nets_amd._lib import lib, check, current_stream
import emd_grid_emulation as EM
L = lib()
rng = np.random.default_rng(5)
n, m = 700, 333
a = rng.random((n, 3), dtype=np.float32) - 0.5
b = rng.random((m, 3), dtype=np.float32) - 0.5
mask = int(os.environ.get("DPF_EMD_DEBUG_SLOTS", "0xffffffff"), 0)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
nbytes = L.dpf_approxmatch_workspace_bytes(1, n, m)
ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
out = torch.empty((m, n), device="cuda"); meta = torch.empty((8,), device="cuda")
d2 = ((b.astype(np.float64)[:, None, :] - a.astype(np.float64)[None, :, :]) ** 2).sum(2)
for j in (7, 6, 5, 3, 0, -1):
    check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "dbg")
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    emu, g, T = EM.exponents(a, b, j, mask=mask)
    ref = -(4.0 ** j) * 1.4426950408889634 * d2
    live = ref > -150
    print("j %2d meta %s | device vs float64 (live pairs): max %.3e | device vs emulation (all pairs, same slots): max %.3e rel %.3e | emulation vs float64 %.3e"
          % (j, meta.cpu().numpy()[3:6], np.abs(got - ref)[live].max(), np.abs(got - emu).max(), np.abs(got - emu).max() / max(np.abs(emu).max(), 1e-30),
             np.abs(emu - ref)[live].max()))
