"""Diagnostic: which rows of pass 2 differ between the matrix-core passes and the packed-VALU kernels, per level, and whether the
matrix-core result depends on the workspace's previous contents / repeats.   emd_rows_probe.py B n m seed"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n, m, seed = (int(v) for v in sys.argv[1:5])
a, b = chamfer_inputs(seed, B, n, m)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
def run(on, fill):
    L.dpf_emd_set_matrix_path(on)
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    ws = torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "x")
    torch.cuda.synchronize()
    return ws[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy().copy(), temp[:, :n + m].cpu().numpy().copy()
ref, _ = run(0, 0)
outs = [run(1, f) for f in (0, 0, 0xFF, 0x7B, 0)]
for i, (o, t) in enumerate(outs[1:]):
    print("run", i + 1, "vs run 0: ratio vectors identical:", np.array_equal(o.view(np.int32), outs[0][0].view(np.int32)))
o = outs[0][0]
for lv in range(9):
    for c in range(B):
        d = np.abs(o[lv, c, n:] - ref[lv, c, n:])
        badrows = np.nonzero(d > 1e-4)[0]
        live = int((ref[lv, c, n:] != 0).sum())
        if len(badrows):
            print("level", 7 - lv, "cloud", c, "live rows (VALU ratioR != 0):", live, "bad rows", badrows[:12], "matrix", o[lv, c, n + badrows[:6]], "valu", ref[lv, c, n + badrows[:6]])
