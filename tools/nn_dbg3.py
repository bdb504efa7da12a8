import sys, os, torch, numpy as np
sys.path.insert(0, '.')
sys.argv = ['bench.py']
import bench
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
args = bench.parse()
dev = torch.device('cuda', 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
stack = dec.stack()
stack.run(z, g, "direct", dec.precision, want_lists=False, n_layers=args.layers, want_pointmajor=True)
pm = stack.last_pointmajor
d1, i1, d2, i2 = BK.NNDistance(pm, tgt_pm)
sw = (i1 >> 16) & 0xFFFF; ve = i1 & 0xFFFF
q = lambda t, p: float(t.float().flatten().kthvalue(max(1, int(p * t.numel())))[0])
print("cycles  dma+setup: med %.0f p99 %.0f | sweep: med %.0f p99 %.0f | verify: med %.0f p99 %.0f" % (q(d1, .5), q(d1, .99), q(sw, .5), q(sw, .99), q(ve, .5), q(ve, .99)))
