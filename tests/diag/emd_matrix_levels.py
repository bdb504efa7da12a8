"""Diagnostic: per-level ratio vectors of the approx-EMD deferred path, matrix-core passes against the packed-VALU kernels."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
from oracle.gen_golden import chamfer_inputs
L = lib()
shapes = [tuple(int(v) for v in sys.argv[1:5])] if len(sys.argv) > 4 else [(2, 64, 64, 764), (3, 300, 257, 1000), (2, 2048, 2048, 2748)]
for (B, n, m, seed) in shapes:
    a, b = chamfer_inputs(seed, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = {}
    for on in (0, 1):
        L.dpf_emd_set_matrix_path(on)
        match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
        nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
        ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
        check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "x")
        torch.cuda.synchronize()
        out[on] = ws[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy().copy()
    for lv in range(9):
        rl0, rl1 = out[0][lv][:, :n], out[1][lv][:, :n]
        rr0, rr1 = out[0][lv][:, n:], out[1][lv][:, n:]
        print((B, n, m), "level", 7 - lv, "per cloud: ratioL max abs/max", ["%.2e" % float(np.abs(rl1[c] - rl0[c]).max() / (np.abs(rl0[c]).max() + 1e-30)) for c in range(B)],
              "ratioR max abs", ["%.2e" % float(np.abs(rr1[c] - rr0[c]).max()) for c in range(B)], "(max %.2f)" % float(rr0.max()))
