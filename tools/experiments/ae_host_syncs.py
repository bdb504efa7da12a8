"""Host-side operations of one steady-state autoencoder (or decoder) training step that can block the host or copy from it:
runtime calls (hipMemcpy*, hipStreamSynchronize, hipEventSynchronize) and the ATen ops that imply them (item, index with host
indices, to / copy_ from the CPU).  usage: ae_host_syncs.py [decoder|autoencoder] [cfg2|cfg3]"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dpf_nets_amd import networks as nets, distributed as D
from torch.profiler import profile, ProfilerActivity
kind = sys.argv[1] if len(sys.argv) > 1 else "autoencoder"
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]
args = types.SimpleNamespace(latent=cfg["latent"], points=cfg["points"], encoder="hip")
dev = torch.device("cuda", 0)
params, compute, store, what = bench.build_train_workload(args, 0, dev, cfg["clouds"], 63, kind)
arena = D.GradArena(params)
opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
def step():
    arena.zero_grad(); loss = compute(); loss.backward(); arena.allreduce(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize()
n = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(n): step()
    torch.cuda.synchronize()
watch = ("Memcpy", "Synchronize", "aten::item", "_local_scalar_dense", "aten::index", "aten::nonzero", "aten::to", "aten::_to_copy",
         "aten::copy_", "aten::tensor", "aten::lift", "aten::masked_select", "aten::is_nonzero", "aten::equal", "EventQuery", "hipMalloc", "hipFree")
for e in sorted(prof.key_averages(), key=lambda e: -e.count):
    if any(w in e.key for w in watch):
        print("%-46s %6.1f calls/step   cpu %8.1f us/step" % (e.key[:46], e.count / n, e.cpu_time_total / n))
