"""How tight can the elementwise criterion on `match` be?  Fraction of entries outside rtol*|ref| + atol for several (rtol, atol),
GPU approxmatch (v_exp_f32) against the C oracle (expf), at the shapes of tests/test_gpu_emd.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle import structural as S
from oracle.gen_golden import chamfer_inputs
S.lib()
for (B, n, m) in [(2, 64, 64), (2, 128, 64), (1, 48, 96), (3, 300, 300), (2, 257, 1024), (1, 2048, 2048)]:
    a, b = chamfer_inputs(500 + n + m, B, n, m)
    match, _ = BK.ApproxMatch(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    gm = match.cpu().numpy()
    rm, _ = S.approxmatch(a, b)
    d = np.abs(gm - rm)
    out = []
    for rt, at in [(5e-3, 1e-4), (1e-3, 1e-5), (1e-3, 1e-6), (1e-4, 1e-6), (1e-4, 1e-7), (1e-5, 1e-7)]:
        out.append("(%g,%g): %.2e" % (rt, at, float((d > rt * np.abs(rm) + at).mean())))
    nz = rm > 1e-6
    print((B, n, m), "max abs %.3e" % d.max(), "max rel on entries > 1e-6: %.3e" % float((d[nz] / rm[nz]).max()), " | ".join(out))
