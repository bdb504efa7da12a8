"""Training step of the flow decoder (inverse flow + PointFlowNLL + backward, training.py:37-55) on one GPU:
the HIP training kernels vs the tensor-op path on PyTorch-ROCm.  Usage: train_bench.py [B N n_flows] [--torch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY
from dpf_nets_amd.networks import train_engine

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, N, n_flows = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (32, 2048, 5)
G = 128
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().train()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
tp = torch.from_numpy(tgt).cuda(); tg = torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()

def step(impl):
    dec.zero_grad(set_to_none=True)
    x = tp.clone().requires_grad_(True)
    ps, mus, lvs = dec(x, tg, mode="inverse") if impl != "torch" else dec.forward_torch(x, tg, mode="inverse")
    loss = nll(ps + [x], [pm] + mus, [pl] + lvs)
    loss.backward()
    return loss

impls = ["bf16x6", "bf16x3"] + (["torch"] if "--torch" in sys.argv else [])
for impl in impls:
    if impl != "torch":
        train_engine.TRAIN_PRECISION = impl
    for _ in range(3): step(impl)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K): l = step(impl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-7s B=%d N=%d L=%d: %.2f ms/step  %.3g point-layers/s (fwd+bwd)  loss %.4f" %
          (impl, B, N, 3 * n_flows, dt * 1e3, B * N * 3 * n_flows / dt, float(l)))
