"""Where does the whole-autoencoder training step spend its GPU time outside the decoder stack?  torch.profiler over a few
steady-state steps of bench.py's autoencoder workload: kernels grouped by name, sorted by total time."""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dpf_nets_amd import networks as nets, distributed as D
from torch.profiler import profile, ProfilerActivity
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = bench.CONFIGS[cfgname]
args = types.SimpleNamespace(latent=cfg["latent"], points=cfg["points"], encoder="hip")
dev = torch.device("cuda", 0)
params, compute, store, what = bench.build_train_workload(args, 0, dev, cfg["clouds"], 63, "autoencoder")
arena = D.GradArena(params)
opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
def step():
    arena.zero_grad(); loss = compute(); loss.backward(); arena.allreduce(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize()
nsteps = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(nsteps): step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 and e.key[:4] not in ("aten", "hipG", "hipL", "hipM")]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print("GPU kernel time per step: %.3f ms" % (tot / nsteps / 1e3))
for e in rows[:45]:
    print("%-90s %7.1f calls/step %9.1f us/step" % (e.key[:90], e.count / nsteps, e.device_time_total / nsteps))
