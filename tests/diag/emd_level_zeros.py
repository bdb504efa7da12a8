"""Diagnostic: how much of every annealing level's weight vectors is exactly zero (points the auction has consumed), per
32-point tile -- what a skip of all-zero tiles in the matrix-core passes could save.  bench.py's cfg5 inputs and independent clouds."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from dpf_nets_amd import synthetic as FO
L = lib()
B, N = 4, 8192
tgt, z, _ = FO.synthetic_inputs(0, B, N, 16)
a_np = np.ascontiguousarray(tgt.transpose(0, 2, 1))
rng = np.random.default_rng(1)
cases = {"cfg5 (jittered, permuted copy)": (a_np[:, rng.permutation(N)] + 0.02 * rng.standard_normal((B, N, 3))).astype(np.float32),
         "independent cloud": np.ascontiguousarray(FO.synthetic_inputs(9, B, N, 16)[0].transpose(0, 2, 1))}
for name, b_np in cases.items():
    ta, tb = torch.from_numpy(a_np).cuda(), torch.from_numpy(b_np).cuda()
    n = m = N
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "x")
    torch.cuda.synchronize()
    lv = ws[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy()
    print(name)
    for q in range(9):
        rl, rr = lv[q][:, :n], lv[q][:, n:]
        tz = lambda v: float((v.reshape(B, -1, 32) == 0).all(-1).mean())
        print("  level %2d: ratioL nonzero %.3f (all-zero 32-tiles %.3f)   ratioR nonzero %.3f (all-zero tiles %.3f)" % (
            7 - q, float((rl != 0).mean()), tz(rl), float((rr != 0).mean()), tz(rr)))
