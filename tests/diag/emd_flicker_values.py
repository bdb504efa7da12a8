"""Diagnostic (r06, tools/asm_bisect): the VALUES behind profiles/r06_emd_bisect.txt's differing bytes.  The reproducer (2 clouds, 128 x 64,
seed 692) stopped behind launch K (default 7: level 6's pass 2) R times; prints, per run, level 6's ratioL[30], ratioR[61] and
remainR[61] of cloud 0 as hex + float, and level 7's for reference.   emd_flicker_values.py [K [R]]
With K = a comma list (e.g. 5,6): every run makes one call per K and prints a digest of the whole workspace + temp behind each -- which
launch's OUTPUT is the first to differ while its INPUT (the digest behind the launch before) is the same, inside one process."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
B, n, m, seed = 2, 128, 64, 692
KS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [7]
K = KS[-1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
A, Bc = chamfer_inputs(seed, B, n, m)
tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(Bc).cuda()
nb = L.dpf_approxmatch_workspace_bytes(B, n, m)
match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
os.environ["DPF_EMD_STOP_AFTER"] = str(K)
lstride = B * (n + m)
def h(x):
    return "%08x %-14.8g" % (np.float32(x).view(np.uint32), x)
import hashlib
for r in range(R if len(KS) > 1 else 0):
    d = []
    for k in KS:
        os.environ["DPF_EMD_STOP_AFTER"] = str(k)
        ws.zero_(); temp.zero_(); match.zero_()
        check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nb, current_stream()), "x")
        torch.cuda.synchronize()
        d.append("K=%d %s" % (k, hashlib.sha1(ws.cpu().numpy().tobytes() + temp.cpu().numpy().tobytes()).hexdigest()[:10]))
    print("run %2d  %s" % (r, "  ".join(d)))
for r in range(R if len(KS) == 1 else 0):
    ws.zero_(); temp.zero_(); match.zero_()
    check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nb, current_stream()), "x")
    torch.cuda.synchronize()
    w = ws.cpu().numpy().view(np.float32)
    t = temp.cpu().numpy().reshape(B, -1)
    l6, l7 = w[lstride:2 * lstride], w[:lstride]
    print("run %2d  L6 ratioL[30] %s ratioR[61] %s remainR[61] %s remainL[30] %s | L7 ratioL[30] %s ratioR[61] %s" % (
        r, h(l6[30]), h(l6[n + 61]), h(t[0, n + 61]), h(t[0, 30]), h(l7[30]), h(l7[n + 61])))
