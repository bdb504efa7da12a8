"""Loss sequence of N optimizer steps at the bench shape (B=32, N=2048, n_flows=21) under graph replay and under eager
launches (DPF_TRAIN_GRAPH=1 / 0), each in its own process: every loss and every gradient norm must agree bit for bit.
usage: replay_vs_eager.py [steps] [nflows]      (the r03 bisection of the drift is recorded in profiles/r03_replay_vs_eager.txt)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from dpf_nets_amd import networks as nets, synthetic as SY
from dpf_nets_amd._lib import lib
steps, nf = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
dec = nets.LocalCondRNVPDecoder(nf, 64, 128).cuda().train()
store = dec.flatten_parameters()
opt = nets.Adam(list(dec.parameters()), lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
tgt, _, g = SY.synthetic_inputs(3, 32, 2048, 128)
tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
pm, pl = torch.zeros(32, 3, 2048).cuda(), torch.full((32, 3, 2048), -3.6).cuda()
nll = nets.PointFlowNLL()
out, gn = [], []
for s in range(steps):
    opt.zero_grad(set_to_none=True)
    ps, mus, lvs = dec(tp, tg, mode="inverse")
    loss = nll(ps + [tp], [pm] + mus, [pl] + lvs)
    loss.backward()
    gn.append(store.flat_g.double().norm().detach())
    opt.step()
    out.append(loss.detach())
print("REPLAYS", lib().dpf_train_graph_replays())
print("LOSS", " ".join(np.float32(x.item()).tobytes().hex() for x in out))
print("GN", " ".join(np.float64(x.item()).tobytes().hex() for x in gn))
print("LAST", out[-1].item())
""" % ROOT
steps = sys.argv[1] if len(sys.argv) > 1 else "50"
nf = sys.argv[2] if len(sys.argv) > 2 else "21"
res = {}
for gr in (0, 1):
    env = dict(os.environ, DPF_TRAIN_GRAPH=str(gr))
    r = subprocess.run([sys.executable, "-c", CHILD, steps, nf], capture_output=True, text=True, env=env)
    if r.returncode:
        print(gr, "FAILED", r.stderr[-600:])
        sys.exit(1)
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.strip().splitlines() if l.strip()}
    res[gr] = lines
    print("graph=%d replays=%s loss after step %s = %s" % (gr, lines["REPLAYS"][0], steps, lines["LAST"][0]), flush=True)
dl = next((i for i, (a, b) in enumerate(zip(res[0]["LOSS"], res[1]["LOSS"])) if a != b), None)
dg = next((i for i, (a, b) in enumerate(zip(res[0]["GN"], res[1]["GN"])) if a != b), None)
print("replay vs eager: first loss difference at step %s, first gradient-norm difference at step %s" % (dl, dg))
sys.exit(0 if dl is None and dg is None else 1)
