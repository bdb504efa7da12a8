"""cProfile of the host side of a training-mode prior-flow step (HIP node)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpf_nets_amd import networks as nets
dec = nets.GlobalRNVPDecoder(7, 128, 128, weight_std=0.05).cuda().train()
g = torch.randn(32, 128, device="cuda")
def step():
    dec.zero_grad(set_to_none=True)
    gin = g.clone().requires_grad_(True)
    gs, mus, lvs = dec(gin, mode="inverse")
    (gs[0].square().mean() + sum(lvs).mean()).backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host issue %.0f us/step, drained after %.0f us more" % ((t1 - t0) / 20 * 1e6, (t2 - t1) * 1e6))
def fwd_only():
    with torch.no_grad():
        pass
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
