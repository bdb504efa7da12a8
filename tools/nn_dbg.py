import sys, torch, numpy as np
sys.path.insert(0, '.')
sys.argv = ['bench.py']
import bench
args = bench.parse()
dev = torch.device('cuda', 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
step = bench.make_step(dec, z, g, tgt_pm, args.layers)
p_out, d1, i1, d2, i2, cd = step()
pm = p_out.transpose(1, 2)
r2 = (pm ** 2).sum(-1)
print("pred |p|^2: max per cloud (first 8)", r2.amax(1)[:8].tolist(), "median", float(r2.median()), "p99", float(r2.flatten().kthvalue(int(0.99 * r2.numel()))[0]))
print("target |p|^2 max", float((tgt_pm ** 2).sum(-1).max()))
print("NN d1 median", float(d1.median()), "d2 median", float(d2.median()), "d1 p99", float(d1.flatten().kthvalue(int(0.99 * d1.numel()))[0]))
