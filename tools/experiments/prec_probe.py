"""flow_kernel duration at a given precision: prec_probe.py <precision> (run through tools/lib_run.py for an alternative build)"""
import sys, torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
args = bench.parse(["--no-extra", "--no-cpu-baseline", "--precision", prec])
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
print(prec, bench.kernel_timings(dec, z, g, tgt_pm, 14, prec))
