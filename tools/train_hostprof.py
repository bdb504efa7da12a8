"""cProfile of the host side of a training step (forward + loss + backward + Adam step).
env: FLAT=1 flatten_parameters(), FUSED=1 Adam(fused=True), NOOPT=1 no optimizer step, TOP=n profile rows."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY  # noqa: E402

B, N, n_flows, G = 32, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 21, 128
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().train()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
tp = torch.from_numpy(tgt).cuda(); tg = torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()
if os.environ.get("FLAT") == "1":
    dec.flatten_parameters()
opt = torch.optim.Adam(dec.parameters(), lr=1e-4, **({"fused": True} if os.environ.get("FUSED") == "1" else {}))
NOOPT = os.environ.get("NOOPT") == "1"


def step():
    opt.zero_grad(set_to_none=True)
    ps, mus, lvs = dec(tp, tg, mode="inverse")
    loss = nll(ps + [tp], [pm] + mus, [pl] + lvs)
    loss.backward()
    if not NOOPT:
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("step %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(int(os.environ.get("TOP", "45")))
