"""Latent prior flow (GlobalRNVPDecoder) in TRAINING mode, forward + backward through csrc/gprior_train.hip.
(r02: replaying its 112 launches as a graph, as the decoder stack and the encoder do, changed nothing -- 2.6 vs 2.4 ms:
the call is bound by the Python side of its autograd node, not by the launches -- so it stays eager.)
usage: python tools/gprior_train_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dpf_nets_amd import networks as nets
    torch.manual_seed(0)
    n_flows, nf, G, B = 7, 128, 128, 32
    prior = nets.GlobalRNVPDecoder(n_flows, nf, G).cuda().train()
    g = torch.randn(B, G, device="cuda", requires_grad=True)

    def step():
        for p in prior.parameters():
            p.grad = None
        gs, mus, lvs = prior(g, mode="inverse")
        (gs[0].square().mean() + sum(lvs).mean()).backward()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    print("%.3f ms per forward + backward (n_flows=%d, nf=%d, G=%d, B=%d)" % ((time.perf_counter() - t0) / 100 * 1e3, n_flows, nf, G, B))


if __name__ == "__main__":
    main()
