"""Host time against GPU time of the autoencoder (or decoder) training step: each phase's HOST duration measured with the device
idle before it (so it is enqueue work, not waiting), beside the step's steady-state wall time.  usage: [decoder|autoencoder] [cfg]"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dpf_nets_amd import networks as nets, distributed as D
kind = sys.argv[1] if len(sys.argv) > 1 else "autoencoder"
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]
args = types.SimpleNamespace(latent=cfg["latent"], points=cfg["points"], encoder="hip")
dev = torch.device("cuda", 0)
params, compute, store, what = bench.build_train_workload(args, 0, dev, cfg["clouds"], 63, kind)
arena = D.GradArena(params)
opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
def step():
    arena.zero_grad(); loss = compute(); loss.backward(); arena.allreduce(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 50
ph = [0.0] * 4
for _ in range(20):
    torch.cuda.synchronize(); a = time.perf_counter(); arena.zero_grad(); loss = compute(); b = time.perf_counter()
    torch.cuda.synchronize(); c = time.perf_counter(); loss.backward(); d = time.perf_counter()
    torch.cuda.synchronize(); e = time.perf_counter(); arena.allreduce(); opt.step(); f = time.perf_counter()
    ph[0] += b - a; ph[1] += d - c; ph[2] += f - e
print("%s: steady-state step %.3f ms;  host enqueue time: forward %.3f  backward %.3f  exchange + optimizer %.3f  = %.3f ms" %
      (kind, wall * 1e3, ph[0] / 20 * 1e3, ph[1] / 20 * 1e3, ph[2] / 20 * 1e3, sum(ph[:3]) / 20 * 1e3))
