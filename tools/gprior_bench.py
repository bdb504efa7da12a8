"""Latent prior flow (GlobalRNVPDecoder, eval mode): the one-launch HIP kernel vs the tensor-op path on the same GPU.
usage: python tools/gprior_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def gpu_time(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    a.record()
    for _ in range(n):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for n_flows, nf, G, B in ((7, 128, 128, 32), (7, 128, 128, 64), (7, 128, 512, 64), (7, 128, 512, 512)):
    torch.manual_seed(0)
    dec = nets.GlobalRNVPDecoder(n_flows, nf, G, weight_std=0.05).cuda().eval()
    g = torch.randn(B, G, device="cuda")
    for mode in ("direct", "inverse"):
        with torch.no_grad():
            fused = timeit(lambda: dec(g, mode=mode))
            kern = gpu_time(lambda: dec.stack().run(g, mode))
            tops = timeit(lambda: dec.forward_torch(g, mode), 10)
        print("n_flows=%d nf=%d G=%d B=%d %-7s  fused %.0f us per call (kernel %.1f us)   tensor ops %.0f us" %
              (n_flows, nf, G, B, mode, fused, kern, tops))

# ---- training mode: forward + loss + backward, HIP node vs tensor ops
for n_flows, nf, G, B in ((7, 128, 128, 32), (7, 128, 512, 64)):
    torch.manual_seed(0)
    dec = nets.GlobalRNVPDecoder(n_flows, nf, G, weight_std=0.05).cuda().train()
    g = torch.randn(B, G, device="cuda")

    def step(fn):
        dec.zero_grad(set_to_none=True)
        gin = g.clone().requires_grad_(True)
        gs, mus, lvs = fn(gin)
        (gs[0].square().mean() + sum(lvs).mean()).backward()

    hip = timeit(lambda: step(lambda x: dec(x, mode="inverse")), 20)
    tops = timeit(lambda: step(lambda x: dec.forward_torch(x, "inverse")), 10)
    store = dec.flatten_parameters()

    def flat_step():
        store.zero_grad()
        gin = g.clone().requires_grad_(True)
        gs, mus, lvs = dec(gin, mode="inverse")
        (gs[0].square().mean() + sum(lvs).mean()).backward()

    flat = timeit(flat_step, 20)
    print("train n_flows=%d nf=%d G=%d B=%d  zero_grad+forward+loss+backward: HIP node %.0f us, with the flat store %.0f us   tensor ops %.0f us"
          % (n_flows, nf, G, B, hip, flat, tops))
