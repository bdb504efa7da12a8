"""Diagnostic: one case of emd_matrix_fuzz.py (seed, index): per level, the ratio vectors of both kernel families against the auction
re-done on the host in float64 from the DEVICE's own exponents (cloud 0).   emd_case_levels.py seed index [cloud]"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib, check, current_stream
L = lib()
seed, index = int(sys.argv[1]), int(sys.argv[2])
cloud = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if os.environ.get("EMD_CASE_GENERATOR") == "suite":      # tests/test_gpu_emd.py::emd_fuzz_case (tools/emd_oracle_fuzz.py's cases)
    from tests.test_gpu_emd import emd_fuzz_case
    rng = np.random.default_rng(seed)
    for it in range(index + 1):
        a, b, kind = emd_fuzz_case(rng)
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
else:
    rng = np.random.default_rng(seed)
    for it in range(index + 1):
        B = int(rng.integers(1, 5))
        n = int(rng.choice([1, 3, 31, 32, 33, 64, 100, 127, 128, 129, 255, 300, 500, 777, 1024, 1500, 2048, 3000]))
        m = int(rng.choice([1, 2, 32, 33, 63, 96, 128, 130, 257, 400, 512, 900, 1024, 2048, 2500]))
        kind = rng.choice(["uniform", "gauss", "jitter", "clustered", "offset", "line"])
        a = rng.random((B, n, 3), dtype=np.float32) - 0.5
        if kind == "gauss": a = (0.2 * rng.standard_normal((B, n, 3))).astype(np.float32)
        if kind == "clustered": a = (a * 0.05 + rng.integers(0, 3, (B, n, 1)) * 0.3).astype(np.float32)
        if kind == "line": a[:, :, 1:] = 0
        if kind == "jitter":
            idx = rng.integers(0, n, m)
            b = (a[:, idx] + 0.02 * rng.standard_normal((B, m, 3))).astype(np.float32)
        else:
            b = (rng.random((B, m, 3), dtype=np.float32) - 0.5) if kind != "gauss" else (0.2 * rng.standard_normal((B, m, 3))).astype(np.float32)
            if kind == "line": b[:, :, 1:] = 0
        if kind == "offset": a, b = a + 5.0, b + 5.0
a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
print("case", index, B, n, m, kind)
tA, tB = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
nbB = L.dpf_approxmatch_workspace_bytes(B, n, m)
def run(on):
    L.dpf_emd_set_matrix_path(on)
    match = torch.empty((B, m, n), device="cuda"); temp = torch.empty((B, (n + m) * 2), device="cuda")
    ws = torch.zeros((nbB,), dtype=torch.uint8, device="cuda")
    check(L.dpf_approxmatch_ws(B, n, m, tA.data_ptr(), tB.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(), nbB, current_stream()), "x")
    torch.cuda.synchronize()
    return ws[:9 * B * (n + m) * 4].view(torch.float32).view(9, B, n + m).cpu().numpy().copy()
mx, va = run(1), run(0)
c = cloud
a0, b0 = np.ascontiguousarray(a[c]), np.ascontiguousarray(b[c])
ta, tb = torch.from_numpy(a0).cuda(), torch.from_numpy(b0).cuda()
nb1 = L.dpf_approxmatch_workspace_bytes(1, n, m)
ws = torch.zeros((nb1,), dtype=torch.uint8, device="cuda")
out = torch.empty((m, n), device="cuda"); meta = torch.empty((8,), device="cuda")
remL = np.full(n, 1.0 if n >= m else float(m // n)); remR = np.full(m, float(n // m) if n >= m else 1.0)
for li, j in enumerate(range(7, -2, -1)):
    check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nb1, current_stream()), "dbg")
    torch.cuda.synchronize()
    w = np.exp2(out.cpu().numpy().astype(np.float64))
    ratioL = remL / (1e-9 + remR @ w); sumr = (w @ ratioL) * remR
    ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR
    remRn = np.maximum(0.0, remR - sumr)
    remLn = np.maximum(0.0, remL - ratioL * (ratioR @ w))
    for name, dev in (("matrix", mx), ("valu  ", va)):
        eL = np.abs(dev[li, c, :n] - ratioL) / (np.abs(ratioL) + 1e-3); eR = np.abs(dev[li, c, n:] - ratioR)
        print("level %2d %s: ratioL worst rel (floor 1e-3) %.2e at %d [dev %.6g host %.6g remL %.3g], ratioR worst abs %.2e at %d" % (
            j, name, eL.max(), int(eL.argmax()), dev[li, c, int(eL.argmax())], ratioL[int(eL.argmax())], remL[int(eL.argmax())], eR.max(), int(eR.argmax())))
    remL, remR = remLn, remRn
