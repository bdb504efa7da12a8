"""Kernel times of the training-mode encoder for an ablated build of its per-point GEMM (make -C dpf_nets_amd/csrc et_ablate ABLATE=<mask>):
et_ab_run.py <so-name> -- prints the average duration of every et_pgemm instantiation (HIP events around each launch are not
available from Python, so the whole forward + backward is timed and rocprofv3 gives the split: run this under
rocprofv3 --kernel-trace --stats, or read the total)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import _lib
so = sys.argv[1]
_lib.lib_path = lambda: os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dpf_nets_amd", so)
import torch
from dpf_nets_amd import networks
torch.manual_seed(0)
B, N = 32, 2048
enc = networks.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda().train()
enc.train_precision = sys.argv[2] if len(sys.argv) > 2 else "bf16x6"
x = (torch.rand(B, 3, N, device="cuda") - 0.5) * 0.5
r = torch.randn(B, 512, device="cuda")
def step():
    for p in enc.parameters():
        p.grad = None
    out = torch.max(enc(x), dim=2)[0]
    (out * r).sum().backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(so, "fwd+bwd ms", (time.perf_counter() - t0) / 20 * 1e3)
