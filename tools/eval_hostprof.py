"""Host side of the eager evaluation step (film + flow + Chamfer through the Python mirror): issue time per step against the
step time, and a cProfile of 500 steps.  r02: 46.6 us of host work per 75.6 us step -- the eager loop is GPU-bound too."""
import sys, time, cProfile, pstats, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse(["--no-extra", "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
step = bench.make_step(dec, z, g, tgt_pm, 14)
for _ in range(200): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host issue %.1f us/step, incl. drain %.1f us/step" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
