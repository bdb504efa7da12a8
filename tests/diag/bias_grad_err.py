"""Relative error (vs float64 autograd) of the cancelling bias gradients of the training stack, per build of the library.
usage: bias_grad_err.py <so> [<so> ...]"""
import os
import subprocess
import sys

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
res = {}
for impl in ('hip', 'torch', 'torch64'):
    dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().train()
    tp = torch.from_numpy(z.copy()).cuda(); tg = torch.from_numpy(g.copy()).cuda()
    if impl == 'torch64': dec, tp, tg = dec.double(), tp.double(), tg.double()
    tp.requires_grad_(True); tg.requires_grad_(True)
    ps, mus, lvs = dec(tp, tg, mode=mode) if impl == 'hip' else dec.forward_torch(tp, tg, mode=mode)
    pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
    loss = nets.PointFlowNLL()([tp] + ps, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()
    loss.backward()
    res[impl] = {k: v.grad.double().cpu().numpy() for k, v in dec.named_parameters() if v.grad is not None}
rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
rows = sorted(((rel(res['hip'][k], res['torch64'][k]), rel(res['torch'][k], res['torch64'][k]), k) for k in res['torch64']), reverse=True)
print(so, 'worst parameter gradients: hip vs f64 | fp32 tensor ops vs f64')
for r in rows[:6]: print('   %%.3e  %%.3e  %%s' %% r)
print('   median over parameters: %%.3e | %%.3e' %% (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
""" % (ROOT, ROOT)
for so in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AB_SO=so), capture_output=True, text=True)
    print(r.stdout.strip() if r.returncode == 0 else r.stderr[-600:])
