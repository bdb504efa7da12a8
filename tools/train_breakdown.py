"""Where a training step's wall time goes (synchronising between the phases)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY
B, N, n_flows, G = 32, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 21, 128
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().train()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
tp = torch.from_numpy(tgt).cuda(); tg = torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = [0, 0, 0, 0]
for it in range(8):
    dec.zero_grad(set_to_none=True)
    x = tp.clone().requires_grad_(True)
    t0 = sync()
    ps, mus, lvs = dec(x, tg, mode="inverse")
    t1h = time.perf_counter(); t1 = sync()
    loss = nll(ps + [x], [pm] + mus, [pl] + lvs)
    t2 = sync()
    loss.backward()
    t3h = time.perf_counter(); t3 = sync()
    if it >= 3:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += (t1h - t0)
    last = (t1h - t0, t1 - t0, t3h - t2, t3 - t2)
print("L=%d  forward %.2f ms (host-side issue %.2f)  loss %.2f ms  backward %.2f ms" % (3 * n_flows, acc[0] / 5 * 1e3, acc[3] / 5 * 1e3, acc[1] / 5 * 1e3, acc[2] / 5 * 1e3))
print("last: fwd host %.2f total %.2f | bwd host %.2f total %.2f" % tuple(v * 1e3 for v in last))
