"""Host time (perf_counter, no synchronisation) of each call of a training step of the flattened 63-layer decoder."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY  # noqa: E402

B, N, G = 32, 2048, 128
torch.manual_seed(0)
dec = nets.LocalCondRNVPDecoder(21, 64, G).cuda().train()
store = dec.flatten_parameters()
opt = nets.Adam(list(dec.parameters()), lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
tgt, _, g = SY.synthetic_inputs(3, B, N, G)
tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()
names = ["zero_grad", "forward", "loss", "backward", "opt.step"]
T = []
for s in range(40):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    ps, mus, lvs = dec(tp, tg, mode="inverse"); t.append(time.perf_counter())
    loss = nll(ps + [tp], [pm] + mus, [pl] + lvs); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if s >= 15:
        T.append(np.diff(t))
    if s == 14:
        torch.cuda.synchronize()
torch.cuda.synchronize()
T = np.array(T) * 1e3
print("host ms per call (median over 25 steps):", {n: round(float(v), 3) for n, v in zip(names, np.median(T, 0))}, "sum %.3f" % np.median(T.sum(1)))
