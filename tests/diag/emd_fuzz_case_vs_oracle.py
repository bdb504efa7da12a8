"""Diagnostic: one case of tests/diag/emd_matrix_fuzz.py (seed, index) against the CPU oracle, both kernel families.
emd_fuzz_case_vs_oracle.py seed index"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle import structural as S
seed, index = int(sys.argv[1]), int(sys.argv[2])
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
print("case", index, B, n, m, kind)
a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
lib().dpf_emd_set_matrix_path(0); _, _, c0 = BK.ApproxMatchCost(ta, tb)
lib().dpf_emd_set_matrix_path(1); _, _, c1 = BK.ApproxMatchCost(ta, tb)
rm, _ = S.approxmatch(a, b); rc = S.matchcost(a, b, rm)
# the same auction in float64 (numpy): how far is the fp32 ORACLE itself from exact arithmetic on this input?
def f64(a1, b1):
    n, m = len(a1), len(b1)
    d2 = ((b1[:, None, :].astype(np.float64) - a1[None, :, :].astype(np.float64)) ** 2).sum(2)
    remL = np.full(n, 1.0 if n >= m else float(m // n)); remR = np.full(m, float(n // m) if n >= m else 1.0); match = np.zeros((m, n))
    for j in range(7, -2, -1):
        e = np.exp(-(4.0 ** j) * d2); ratioL = remL / (1e-9 + remR @ e); sumr = (e @ ratioL) * remR
        ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR; remR = np.maximum(0.0, remR - sumr)
        w = e * ratioR[:, None] * ratioL[None, :]; match += w; remL = np.maximum(0.0, remL - w.sum(0))
    return float((match * np.sqrt(d2)).sum())
r64 = np.array([f64(a[i], b[i]) for i in range(B)])
print("oracle (fp32 C)", rc, " float64 auction", r64, " oracle vs float64", np.abs(rc - r64) / r64)
print("packed VALU   ", c0.cpu().numpy(), "rel to oracle", np.abs(c0.cpu().numpy() - rc) / rc, "rel to float64", np.abs(c0.cpu().numpy() - r64) / r64)
print("matrix cores  ", c1.cpu().numpy(), "rel to oracle", np.abs(c1.cpu().numpy() - rc) / rc, "rel to float64", np.abs(c1.cpu().numpy() - r64) / r64)

# ---- the passes' exponents of cloud 0 against float64, level by level, and the auction re-done on the host from them
from dpf_nets_amd._lib import check, current_stream
L = lib()
a0, b0 = np.ascontiguousarray(a[0]), np.ascontiguousarray(b[0])
t0a, t0b = torch.from_numpy(a0).cuda(), torch.from_numpy(b0).cuda()
nb1 = L.dpf_approxmatch_workspace_bytes(1, n, m)
ws = torch.zeros((nb1,), dtype=torch.uint8, device="cuda")
out = torch.empty((m, n), device="cuda"); meta = torch.empty((8,), device="cuda")
d2 = ((b0.astype(np.float64)[:, None, :] - a0.astype(np.float64)[None, :, :]) ** 2).sum(2)
remL = np.full(n, 1.0 if n >= m else float(m // n)); remR = np.full(m, float(n // m) if n >= m else 1.0); match = np.zeros((m, n))
for j in range(7, -2, -1):
    check(L.dpf_debug_emd_exponents(n, m, t0a.data_ptr(), t0b.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(), nb1, current_stream()), "dbg")
    torch.cuda.synchronize()
    e = out.cpu().numpy().astype(np.float64); ref = -(4.0 ** j) * 1.4426950408889634 * d2
    live = ref > -150
    print("level %2d meta %s exponent error on live pairs: max %.2e (relative to |e| max %.2e)" % (j, meta.cpu().numpy()[3:6], np.abs(e - ref)[live].max(), (np.abs(e - ref)[live] / np.maximum(np.abs(ref[live]), 1e-9)).max()))
    w = np.exp2(e)
    ratioL = remL / (1e-9 + remR @ w); sumr = (w @ ratioL) * remR
    ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR; remR = np.maximum(0.0, remR - sumr)
    ww = w * ratioR[:, None] * ratioL[None, :]; match += ww; remL = np.maximum(0.0, remL - ww.sum(0))
print("float64 auction on the DEVICE's exponents: cost", float((match * np.sqrt(d2)).sum()), "vs float64 on exact exponents", r64[0])
