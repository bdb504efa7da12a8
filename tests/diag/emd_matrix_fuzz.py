"""Diagnostic: random shapes (ragged, n != m, tiny, large ratios) and cloud kinds through the matrix-core approx-EMD against the
packed-VALU kernels: cost, matching, mass, NaNs, and bit-stability of a second call.   emd_matrix_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = dict(cost=0.0, match=0.0, mass=0.0)
rows = []
for it in range(cases):
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
    ta, tb = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(b, np.float32)).cuda()
    lib().dpf_emd_set_matrix_path(0)
    m0, t0, c0 = BK.ApproxMatchCost(ta, tb)
    lib().dpf_emd_set_matrix_path(1)
    m1, t1, c1 = BK.ApproxMatchCost(ta, tb)
    m2, t2, c2 = BK.ApproxMatchCost(ta, tb)
    torch.cuda.synchronize()
    assert torch.equal(m1, m2) and torch.equal(c1, c2), ("not repeatable", B, n, m, kind)
    assert torch.isfinite(m1).all() and torch.isfinite(c1).all(), ("non-finite", B, n, m, kind)
    ce = float(((c1 - c0).abs() / (c0.abs() + 1e-12)).max()); me = float((m1 - m0).abs().max())
    ma = float(max((m1.sum(1) - m0.sum(1)).abs().max(), (m1.sum(2) - m0.sum(2)).abs().max()))
    worst = dict(cost=max(worst["cost"], ce), match=max(worst["match"], me), mass=max(worst["mass"], ma))
    rows.append((ce, me, ma, it, B, n, m, str(kind), float(c0.abs().min())))
    flag = "" if (ce <= 1e-5 and me <= 2e-3 and ma <= 1e-4) else "   <-- outside the test bars"
    print("%3d B=%d n=%4d m=%4d %-9s cost rel %.1e  match %.1e  mass %.1e%s" % (it, B, n, m, kind, ce, me, ma, flag))
print("worst:", worst)
for r in sorted(rows, reverse=True)[:6]:
    print("largest cost difference: %.2e (match %.1e, mass %.1e) case %d B=%d n=%d m=%d %s, smallest cost in the batch %.3g" % r)
