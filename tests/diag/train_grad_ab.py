"""Parameter / input gradients of the training stack from two builds of the library, elementwise (diagnosis).
usage: train_grad_ab.py <so A> <so B>"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from dpf_nets_amd import _lib
so = os.environ['AB_SO']
_lib.lib_path = lambda: os.path.join(%r, 'dpf_nets_amd', so)
from dpf_nets_amd import networks as nets, synthetic as FO
B, N, mode = 8, 2048, 'direct'
n_flows, G, seed = 2, 128, 31
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in FO.make_decoder_state(seed, n_flows, 64, G).items()}
tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
dec.load_state_dict(sd, strict=True)
dec = dec.cuda().train()
tp = torch.from_numpy(z.copy()).cuda().requires_grad_(True)
tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
ps, mus, lvs = dec(tp, tg, mode=mode)
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
loss = nets.PointFlowNLL()([tp] + ps, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()
loss.backward()
out = {'gp': tp.grad.cpu().numpy(), 'gg': tg.grad.cpu().numpy(), 'loss': np.float64(float(loss))}
for k, v in dec.named_parameters():
    if v.grad is not None:
        out['g/' + k] = v.grad.cpu().numpy()
np.savez(os.environ['AB_OUT'], **out)
""" % (ROOT, ROOT)
res = []
for i, so in enumerate(sys.argv[1:3]):
    env = dict(os.environ, AB_SO=so, AB_OUT="/tmp/ab_%d.npz" % i)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if r.returncode:
        print(so, r.stderr[-800:])
        sys.exit(1)
    res.append(np.load("/tmp/ab_%d.npz" % i))
a, b = res
rows = []
for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k]).max()
    rows.append((d / (np.abs(a[k]).max() + 1e-30), k, d))
rows.sort(reverse=True)
print("identical:", sum(1 for r in rows if r[2] == 0), "of", len(rows))
for r in rows[:12]:
    print("%.3e  %s  (abs %.3e)" % r)
