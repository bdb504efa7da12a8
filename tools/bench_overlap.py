"""Experiment: consecutive steps of bench.py's workload on two streams (two captured graphs with their own buffers), so that
the Chamfer kernels of step i can share the chip with the flow kernel of step i+1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
device = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, device, 32)
step = bench.make_step(dec, z, g, tgt_pm, args.layers)
for _ in range(3):
    step()
torch.cuda.synchronize()
NS = int(os.environ.get('NS', '4'))
streams = [torch.cuda.Stream() for _ in range(NS)]
graphs = []
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        step()
    graphs.append(gr)
torch.cuda.synchronize()


def run(n, ns):
    for i in range(n):
        k = i % ns
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()


for ns in (1, 2, 3, 4, 1, 2, 3, 4):
    run(20, ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(400, ns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    print("%d stream(s): %.1f us/step  %.3e points/s" % (ns, dt * 1e6, 32 * args.points / dt))
