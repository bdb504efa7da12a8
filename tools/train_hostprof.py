"""cProfile of the host side of a training step (forward + loss + backward + Adam step).
env: FLAT=1 flatten_parameters(), OPT=torch|mirror|loop (torch.optim.Adam | networks.optimizers.Adam | the reference's
per-parameter loop), FUSED=1 torch Adam(fused=True), NOOPT=1 no optimizer step, TOP=n profile rows."""
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
class LoopAdam(torch.optim.Optimizer):
    """The per-parameter Python loop the reference's optimizer is (lib/networks/optimizers.py:19-74), for timing."""

    def __init__(self, params, lr):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self):
        import math
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0; st["m"] = torch.zeros_like(p); st["v"] = torch.zeros_like(p); st["vmax"] = torch.zeros_like(p)
                st["step"] += 1
                st["m"].mul_(0.9).add_(p.grad, alpha=0.1)
                st["v"].mul_(0.995).addcmul_(p.grad, p.grad, value=0.005)
                torch.max(st["vmax"], st["v"], out=st["vmax"])
                denom = st["vmax"].sqrt()
                mc = st["m"] / (1 - 0.9 ** st["step"])
                dc = torch.add(denom / math.sqrt(1 - 0.995 ** st["step"]), 1e-8)
                p.add_(-torch.addcdiv(torch.mul(p, 1e-6), mc, dc, value=group["lr"]))


OPT = os.environ.get("OPT", "torch")
if OPT == "mirror":
    opt = nets.Adam(dec.parameters(), lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.995), amsgrad=True)
elif OPT == "loop":
    opt = LoopAdam(dec.parameters(), lr=1e-4)
else:
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
