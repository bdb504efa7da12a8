"""Diagnostic: dpf_debug_emd_exponents on one cloud of chamfer_inputs(seed, B, n, m): rows whose exponents are off.
emd_exponent_probe2.py B n m seed cloud"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
import emd_grid_emulation as EM
L = lib()
B, n, m, seed, c = (int(v) for v in sys.argv[1:6])
a, b = chamfer_inputs(seed, B, n, m)
a, b = np.ascontiguousarray(a[c]), np.ascontiguousarray(b[c])
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
nbytes = L.dpf_approxmatch_workspace_bytes(1, n, m)
d2 = ((b.astype(np.float64)[:, None, :] - a.astype(np.float64)[None, :, :]) ** 2).sum(2)
for fill in (0, 0xFF):
    ws = torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")
    out = torch.empty((m, n), device="cuda"); meta = torch.empty((8,), device="cuda")
    for j in (7, 6, 5):
        check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "dbg")
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        ref = -(4.0 ** j) * 1.4426950408889634 * d2
        err = np.abs(got - ref)
        live = ref > -150
        rows = np.nonzero((err * live).max(1) > 1e-3)[0]
        print("fill %#x j %d meta %s max err on live pairs %.3e; rows off by > 1e-3: %s" % (fill, j, meta.cpu().numpy()[3:6], (err * live).max(), rows[:10]))
        for r in rows[:2]:
            k = int(np.argmax(err[r] * live[r]))
            print("    row", r, "col", k, "device", got[r, k], "float64", ref[r, k], "point", b[r], "centre", meta.cpu().numpy()[:3])

# ---- pass 2 of level 6 re-done on the host from the DEVICE's exponents (debug kernel) and the device's ratioL of that level
if len(sys.argv) > 6:
    a2, b2 = chamfer_inputs(seed, B, n, m)
    ta2, tb2 = torch.from_numpy(a2).cuda(), torch.from_numpy(b2).cuda()
    nb2 = L.dpf_approxmatch_workspace_bytes(B, n, m)
    L.dpf_emd_set_matrix_path(1)
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    ws2 = torch.zeros((nb2,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, ta2.data_ptr(), tb2.data_ptr(), match.data_ptr(), temp.data_ptr(), ws2.data_ptr(), nb2, current_stream()), "x")
    torch.cuda.synchronize()
    lev = ws2[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy()
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
    remR = np.full(m, float(n // m) if n >= m else 1.0, np.float32)
    for li, j in enumerate((7, 6, 5, 4)):
        check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "dbg")
        torch.cuda.synchronize()
        e = out.cpu().numpy()
        w = np.exp2(e.astype(np.float64))
        ratioL = lev[li, c, :n].astype(np.float64)
        tot = (w * ratioL[None, :]).sum(1)
        sumr = tot * remR
        ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR
        dev = lev[li, c, n:]
        bad = np.nonzero(np.abs(dev - ratioR) > 1e-4 * (1 + np.abs(ratioR)))[0]
        print("level", j, "pass 2 redone from the debug exponents vs the pass's ratioR: rows that differ", bad[:10],
              "device", dev[bad[:4]], "redone", ratioR[bad[:4]], "| any non-finite / positive exponent:", bool((~np.isfinite(e)).any()), float(e.max()))
        remR = np.maximum(0.0, remR - sumr).astype(np.float32)
        remR_dev_next = None
