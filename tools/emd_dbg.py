import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle.gen_golden import chamfer_inputs
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
for (B, n, m) in ((2, 64, 64), (3, 300, 257)):
    a, b = chamfer_inputs(700 + n, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    BK.EMD_RMW = True
    m1, t1 = BK.ApproxMatch(ta, tb)
    BK.EMD_RMW = False
    m2, t2 = BK.ApproxMatch(ta, tb)
    d = (m1 - m2).abs()
    print(B, n, m, "match maxdiff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(),
          "temp diff", float((t1[:, :n+m] - t2[:, :n+m]).abs().max()))
    idx = (d > 0).nonzero()[:5]
    for i in idx: print("  ", i.tolist(), float(m1[tuple(i)]), float(m2[tuple(i)]))
print("determinism check")
a, b = chamfer_inputs(764, 2, 64, 64)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
for mode in (True, False):
    BK.EMD_RMW = mode
    r = [BK.ApproxMatch(ta, tb) for _ in range(3)]
    print(" rmw" if mode else " deferred", [float((r[0][0] - x[0]).abs().max()) for x in r[1:]], [float((r[0][1][:, :128] - x[1][:, :128]).abs().max()) for x in r[1:]])
