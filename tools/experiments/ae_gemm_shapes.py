"""Which tensor-op GEMMs does the autoencoder training step issue, with which shapes and how long do they take?"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dpf_nets_amd import networks as nets, distributed as D
from torch.profiler import profile, ProfilerActivity
cfg = bench.CONFIGS["cfg2"]
args = types.SimpleNamespace(latent=cfg["latent"], points=cfg["points"], encoder="hip")
dev = torch.device("cuda", 0)
params, compute, store, what = bench.build_train_workload(args, 0, dev, cfg["clouds"], 63, "autoencoder")
arena = D.GradArena(params)
opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
def step():
    arena.zero_grad(); loss = compute(); loss.backward(); arena.allreduce(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(4): step()
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::addmm", "aten::mm", "aten::bmm", "aten::linear", "aten::matmul", "aten::baddbmm") and e.device_time_total > 0:
        print("%-14s %-70s calls %3d  device %8.1f us each" % (e.key, str(e.input_shapes)[:70], e.count, e.device_time_total / e.count))
