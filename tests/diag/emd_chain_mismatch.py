"""Diagnostic (EMD_DBG=8 build, tools/lib_run.py libdpf_dbg8.so): every pair_exponents of every matrix-core kernel computes the chained
MFMA pair AND the two MFMAs unchained; lanes where they differ are counted and the first 16 recorded."""
import os, sys, ctypes
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream, lib_path
from oracle.gen_golden import chamfer_inputs
H = ctypes.CDLL(lib_path())
L = lib()
B, n, m, seed = (int(v) for v in sys.argv[1:5])
A, Bc = chamfer_inputs(seed, B, n, m)
tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(Bc).cuda()
nb = L.dpf_approxmatch_workspace_bytes(B, n, m)
for it in range(3):
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    ws = torch.zeros((nb,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nb, current_stream()), "x")
    torch.cuda.synchronize()
    out = (ctypes.c_float * 129)()
    H.dpf_debug_emd_mismatches(out)
    o = np.array(out[:])
    print("call", it, "lanes x registers where chained != unchained:", int(o[0]))
    for r in o[1:].reshape(16, 8)[: min(int(o[0]), 8)]:
        print("   chained %.6g unchained %.6g (= first %.6g + second %.6g)  thread %d reg %d block (%d, %d)" % tuple(r))
