"""DUAL body of the flow stack (csrc/flow_dual.h, DPF_FLOW_DUAL=1) against the default body: same bits expected, and timing.
Runs each variant in a child process (the choice is read once per process)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from dpf_nets_amd import networks as nets, synthetic as SY
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
args = bench.parse()
dev = torch.device('cuda', 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, args.batch or 32)
stack = dec.stack()
L = args.layers
with torch.no_grad():
    p_out, sum_lv, ps, mus, lvs = stack.run(z, g, "direct", dec.precision, want_lists=True, n_layers=L, want_pointmajor=True)
    last = p_out.cpu().numpy(); slv = sum_lv.cpu().numpy(); pm = stack.last_pointmajor.cpu().numpy()
    l3 = ps.cpu().numpy(); m3 = mus.cpu().numpy(); v3 = lvs.cpu().numpy()
    q = stack.run(torch.from_numpy(tgt).to(dev), g, "inverse", dec.precision, want_lists=False, n_layers=L, want_pointmajor=False)
    i0 = q[0].cpu().numpy(); i1 = q[1].cpu().numpy()
np.savez(os.environ['DUAL_OUT'], last=last, slv=slv, pm=pm, l3=l3, m3=m3, v3=v3, i0=i0, i1=i1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from dpf_nets_amd.networks import engine
with torch.no_grad():
    run = lambda: stack.run(z, g, "direct", dec.precision, want_lists=False, n_layers=L, want_pointmajor=True)
    for _ in range(50): run()
    torch.cuda.synchronize(); s.record()
    for _ in range(200): run()
    e.record(); e.synchronize()
print('us per decode (film + flow):', s.elapsed_time(e) / 200 * 1e3)
""" % ROOT


def main():
    outs = {}
    for dual in ("0", "1"):
        env = dict(os.environ, DPF_FLOW_DUAL=dual, DUAL_OUT="/tmp/dual_%s.npz" % dual)
        r = subprocess.run([sys.executable, "-c", CHILD] + sys.argv[1:], env=env, capture_output=True, text=True)
        print("DUAL=%s:" % dual, r.stdout.strip()[-300:], r.stderr.strip()[-600:] if r.returncode else "")
        if r.returncode == 0:
            outs[dual] = np.load("/tmp/dual_%s.npz" % dual)
    if len(outs) == 2:
        for k in outs["0"].files:
            a, b = outs["0"][k], outs["1"][k]
            print("  %-5s max|diff| %.3e  bitwise %s  finite %s" % (k, np.abs(a - b).max(), np.array_equal(a, b), np.isfinite(b).all()))


if __name__ == "__main__":
    main()
