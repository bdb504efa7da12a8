import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle.gen_golden import chamfer_inputs
n = 8192
a, b = chamfer_inputs(4242, 2, n, n)
b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
for it in range(4):
    match, temp, cost = BK.ApproxMatchCost(ta, tb)
    torch.cuda.synchronize()
    s1, s2 = match.sum(1), match.sum(2)
    print(it, "max col sum", float(s1.max()), "max row sum", float(s2.max()), "cost", cost.cpu().numpy(), "n > 1.001:", int((s1 > 1.001).sum()), int((s2 > 1.001).sum()))
    if it == 0:
        w = (s1 > 1.001).nonzero()
        print("  where", w[:5].tolist(), s1[s1 > 1.001][:5].tolist())
