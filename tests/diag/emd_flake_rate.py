"""Diagnostic: how often the first level's passes of the matrix-core approx-EMD return different bits on the same input.
N calls; the first level's ratioL (pass 1) and ratioR (pass 2) of every call against call 0's.  (During the r05 hunt the library
had two debug switches, since removed: DPF_EMD_DBG_STOP -- return behind the first pass, which showed pass 1 alone flickering --
and DPF_EMD_DBG_S -- slices per workgroup, which showed it needs two waves per SIMD.)   FB=<clouds> emd_flake_rate.py [n] [calls]"""
import os, sys, collections
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n = int(os.environ.get("FB", "2")), int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = n
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
a, b = chamfer_inputs(4242, B, n, n)
b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
res = []
for it in range(reps):
    check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "x")
    torch.cuda.synchronize()
    lv0 = ws[:B * (n + m) * 4].view(torch.int32).view(B, n + m).cpu().numpy().copy()
    res.append(lv0)
ref = res[0]
cntL = sum(int((r[:, :n] != ref[:, :n]).any()) for r in res); cntR = sum(int((r[:, n:] != ref[:, n:]).any()) for r in res)
nL = [int((r[:, :n] != ref[:, :n]).sum()) for r in res]; nR = [int((r[:, n:] != ref[:, n:]).sum()) for r in res]
print("n = %d, %d calls: calls whose level-7 ratioL differs from call 0: %d (entries per call: %s)" % (n, reps, cntL, nL))
print("                  calls whose level-7 ratioR differs from call 0: %d (entries per call: %s)" % (cntR, nR))
seen = collections.Counter()
for r in res:
    d = np.argwhere(r != ref)
    for bi, j in d:
        seen[(int(bi), int(j), int(r[bi, j]), int(ref[bi, j]))] += 1
for (bi, j, v, v0), c in seen.most_common(12):
    print("  cloud %d %s %d: %.9g instead of %.9g in %d calls" % (bi, "ratioL k =" if j < n else "ratioR l =", j if j < n else j - n,
          np.int32(v).view(np.float32), np.int32(v0).view(np.float32), c))

