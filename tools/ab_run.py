import os, sys, ctypes
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from dpf_nets_amd import _lib
ab = sys.argv[1]
_lib.lib_path = lambda: os.path.join(os.environ["GRAFT_REPO_ROOT"], "dpf_nets_amd", "libdpf_ab%s.so" % ab if ab != "0" else "libdpf_hip.so")
import bench
args = bench.parse(["--no-extra", "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
kt = bench.kernel_timings(dec, z, g, tgt_pm, 14, "f16x3")
print("ablate=%s flow_kernel %.2f us" % (ab, kt["flow_kernel"]))
