"""Diagnostic: the one case of emd_matrix_fuzz.py (seed 11, case 66: 4 clouds of 1500 collinear points against 128) where the two
GPU kernel families' costs differ by 1e-4 -- which of them is nearer the CPU oracle, and how far is the oracle's own fp32 from a
float64 run of the same auction?"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle import structural as S
rng = np.random.default_rng(11)
for it in range(67):
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
print("case", it, B, n, m, kind)
a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
lib().dpf_emd_set_matrix_path(0); _, _, c0 = BK.ApproxMatchCost(ta, tb)
lib().dpf_emd_set_matrix_path(1); _, _, c1 = BK.ApproxMatchCost(ta, tb)
rm, _ = S.approxmatch(a, b); rc = S.matchcost(a, b, rm)
print("oracle        ", rc)
print("packed VALU   ", c0.cpu().numpy(), "rel to oracle", np.abs(c0.cpu().numpy() - rc) / rc)
print("matrix cores  ", c1.cpu().numpy(), "rel to oracle", np.abs(c1.cpu().numpy() - rc) / rc)
