"""Diagnostic: which launch of the matrix-core approx-EMD family first leaves different bytes behind on two runs of the same input?
DPF_EMD_STOP_AFTER=k makes dpf_approxmatch_ws return behind the family's k-th launch (csrc/emd.hip); for k = 1, 2, ... the call runs
R times on freshly zeroed buffers and the whole workspace and `temp` are compared byte for byte.   emd_bisect.py B n m seed [R [K0 [K1]]]"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n, m, seed = (int(v) for v in sys.argv[1:5])
R = int(sys.argv[5]) if len(sys.argv) > 5 else 6
K0 = int(sys.argv[6]) if len(sys.argv) > 6 else 1        # start at launch K0 (and do not stop at the first difference if given)
K1 = int(sys.argv[7]) if len(sys.argv) > 7 else None     # ... and end behind launch K1
A, Bc = chamfer_inputs(seed, B, n, m)
tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(Bc).cuda()
nb = L.dpf_approxmatch_workspace_bytes(B, n, m)
match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
names = ["pack"]
for j in range(7, -2, -1):
    names += ["L%d pass1" % j, "L%d pass2 (4 tiles)" % j] + (["L%d pass2 (1 tile)" % j] if j < 7 else []) + ["L%d pass3" % j, "L%d compact" % j]
names += ["gather", "materialise"]
for k in range(K0, (K1 if K1 is not None else len(names)) + 1):
    os.environ["DPF_EMD_STOP_AFTER"] = str(k)
    snaps = []
    for r in range(R):
        ws.zero_(); temp.zero_(); match.zero_()
        check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nb, current_stream()), "x")
        torch.cuda.synchronize()
        snaps.append((ws.clone(), temp.clone().view(torch.uint8).flatten(), match.clone().view(torch.uint8).flatten()))
    diff = [sum(int((snaps[r][i] != snaps[0][i]).sum()) for i in range(3)) for r in range(1, R)]
    if any(diff):
        first = next(r for r in range(1, R) if diff[r - 1])
        w = torch.nonzero(snaps[first][0] != snaps[0][0]).flatten()
        t = torch.nonzero(snaps[first][1] != snaps[0][1]).flatten()
        print("launch %d (%s): runs differ (bytes: %s); workspace byte offsets %s ... temp byte offsets %s" % (k, names[k - 1], diff, w[:6].tolist(), t[:6].tolist()))
        if K0 == 1:
            break
        continue
    print("launch %d (%s): %d runs identical" % (k, names[k - 1], R))
