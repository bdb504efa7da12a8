import sys, torch
import bench
args = bench.parse(["--no-extra", "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
print(bench.kernel_timings(dec, z, g, tgt_pm, 14, args.precision))
