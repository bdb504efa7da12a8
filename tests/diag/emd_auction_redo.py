"""Diagnostic: the whole auction of ONE cloud re-done on the host (float64 sums) from the DEVICE's own exp2 arguments
(dpf_debug_emd_exponents), level by level, against the ratio vectors the matrix-core passes and the packed-VALU kernels left in the
workspace: which pass of which level first departs, and on which points.   emd_auction_redo.py B n m seed cloud"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n, m, seed, c = (int(v) for v in sys.argv[1:6])
A, Bc = chamfer_inputs(seed, B, n, m)
tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(Bc).cuda()
nbB = L.dpf_approxmatch_workspace_bytes(B, n, m)
def run(on):
    L.dpf_emd_set_matrix_path(on)
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    ws = torch.zeros((nbB,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbB, current_stream()), "x")
    torch.cuda.synchronize()
    return ws[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy().copy()
mx, va = run(1), run(0)
a, b = np.ascontiguousarray(A[c]), np.ascontiguousarray(Bc[c])
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
nb1 = L.dpf_approxmatch_workspace_bytes(1, n, m)
ws = torch.zeros((nb1,), dtype=torch.uint8, device="cuda")
out = torch.empty((m, n), device="cuda"); meta = torch.empty((8,), device="cuda")
multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
remL, remR = np.full(n, multiL), np.full(m, multiR)
def rel(x, y): return np.abs(x - y) / (np.abs(y) + 1e-6)
for li, j in enumerate(range(7, -2, -1)):
    check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nb1, current_stream()), "dbg")
    torch.cuda.synchronize()
    w = np.exp2(out.cpu().numpy().astype(np.float64))          # (m, n)
    ratioL = remL / (1e-9 + (w * remR[:, None]).sum(0))
    sumr = (w * ratioL[None, :]).sum(1) * remR
    ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR
    remR = np.maximum(0.0, remR - sumr)
    remL = np.maximum(0.0, remL - ratioL * (w * ratioR[:, None]).sum(0))
    for name, dev in (("matrix", mx), ("valu  ", va)):
        eL, eR = rel(dev[li, c, :n], ratioL), np.abs(dev[li, c, n:] - ratioR)
        print("level %2d %s vs host redo: ratioL worst rel %.2e (point %d), ratioR worst abs %.2e (point %d)" % (j, name, eL.max(), int(eL.argmax()), eR.max(), int(eR.argmax())))
