"""Diagnostic: the matrix-core approx-EMD run several times on the same input -- which level's ratio vectors differ first."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n, m = (2, int(sys.argv[1]), int(sys.argv[1])) if len(sys.argv) > 1 else (2, 2048, 2048)
a, b = chamfer_inputs(700 + n, B, n, m)
if len(sys.argv) > 2:
    a, b = chamfer_inputs(4242, 2, n, n); b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
runs = []
for it in range(6):
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "x")
    torch.cuda.synchronize()
    runs.append((ws[:9 * B * (n + m) * 4].view(torch.int32).view(9, B, n + m).cpu().numpy().copy(), match.view(torch.int32).cpu().numpy().copy()))
for it in range(1, 6):
    d = [(int((runs[it][0][q][:, :n] != runs[0][0][q][:, :n]).sum()), int((runs[it][0][q][:, n:] != runs[0][0][q][:, n:]).sum())) for q in range(9)]
    print("run", it, "differing (ratioL, ratioR) per level 7..-1:", d, "match", int((runs[it][1] != runs[0][1]).sum()))
q = 3
for it in (1, 3):
    x0, x1 = runs[0][0][q].view(np.float32), runs[it][0][q].view(np.float32)
    bb, ll = np.nonzero(x0[:, n:] != x1[:, n:])
    live = (runs[0][0][q][:, n:] != 0).sum(1)
    print("level 4, run", it, "live rows per cloud", live)
    for c, l in zip(bb, ll):
        # place of row l in the level's list = number of live rows before it
        place = int((runs[0][0][q][c, n:n + l] != 0).sum())
        print("   cloud", c, "row", l, "place", place, "values", x0[c, n + l], x1[c, n + l])
