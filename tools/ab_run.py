"""Time flow_kernel (HIP events, cfg-2) of an ablation build: `make -C dpf_nets_amd/csrc ablate ABLATE=<mask>` produces
dpf_nets_amd/libdpf_ab<mask>.so with one piece of the kernel's work compiled out (flow.hip, DPF_ABLATE bits; the results of
such a build are garbage -- timing only).  usage: ab_run.py <mask>   (0 = the product library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dpf_nets_amd import _lib  # noqa: E402

ab = sys.argv[1]
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", "libdpf_ab%s.so" % ab if ab != "0" else "libdpf_hip.so")
import bench  # noqa: E402

args = bench.parse(["--no-extra", "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
kt = bench.kernel_timings(dec, z, g, tgt_pm, 14, args.precision)
print("ablate=%s flow_kernel %.2f us" % (ab, kt["flow_kernel"]))
