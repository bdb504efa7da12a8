"""Diagnostic: the approx-EMD deferred path on the matrix cores against the packed-VALU kernels (same call, dpf_emd_set_matrix_path)."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle.gen_golden import chamfer_inputs
for (B, n, m) in ((2, 64, 64), (1, 48, 96), (3, 300, 257), (2, 1024, 2048), (2, 2048, 2048)):
    a, b = chamfer_inputs(700 + n, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    lib().dpf_emd_set_matrix_path(0)
    m0, t0, c0 = BK.ApproxMatchCost(ta, tb)
    lib().dpf_emd_set_matrix_path(1)
    m1, t1, c1 = BK.ApproxMatchCost(ta, tb)
    torch.cuda.synchronize()
    d = (m1 - m0).abs()
    print((B, n, m), "cost", c0.cpu().numpy(), c1.cpu().numpy(), "rel", float(((c1 - c0).abs() / c0).max()),
          "match max|diff|", float(d.max()), "of", float(m0.max()), "remain diff", float((t1[:, :n + m] - t0[:, :n + m]).abs().max()))
