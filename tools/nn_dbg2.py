import sys, os, torch, numpy as np
sys.path.insert(0, '.')
sys.argv = ['bench.py']
import bench
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
args = bench.parse()
dev = torch.device('cuda', 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
stack = dec.stack()
p_out, _, _, _, _ = stack.run(z, g, "direct", dec.precision, want_lists=False, n_layers=args.layers, want_pointmajor=True)
pm = stack.last_pointmajor
d1, i1, d2, i2 = BK.NNDistance(pm, tgt_pm)
print("dir1 (pred queries -> target cands): qcount mean %.2f max %d p99 %d ; overflow waves frac %.4f" % (float(d1.mean()), int(d1.max()), int(d1.flatten().kthvalue(int(0.99*d1.numel()))[0]), float(i1.float().mean())))
print("dir2 (target queries -> pred cands): qcount mean %.2f max %d p99 %d ; overflow waves frac %.4f" % (float(d2.mean()), int(d2.max()), int(d2.flatten().kthvalue(int(0.99*d2.numel()))[0]), float(i2.float().mean())))
if os.environ.get("DPF_NNM_DEBUG") == "2":
    print("cycles: sweep median %.0f p90 %.0f | verify median %.0f p90 %.0f max %d" % (float(d1.median()), float(d1.flatten().kthvalue(int(0.9*d1.numel()))[0]), float(i1.float().median()), float(i1.float().flatten().kthvalue(int(0.9*i1.numel()))[0]), int(i1.max())))
