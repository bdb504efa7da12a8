"""Diagnostic: over many seeds at (8, 2048, direct), default precision -- the error of d loss / d g (a row sums over ONE cloud's
N points) of the HIP training stack AND of the fp32 tensor-op path, both against float64, in units of 1 / N; and the worst parameter
gradient excess in units of 1 / P.  Shows what a ReLU that falls the other way than in float64 costs on either path.
    gg_seeds.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import flow_oracle as FO
from dpf_nets_amd import networks as nets

def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().cpu().numpy().astype(np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

B, N, mode, n_flows, G = 8, 2048, "direct", 2, 128
first = int(sys.argv[1]) if len(sys.argv) > 1 else 31
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = []
for seed in range(first, first + count):
    sd = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
    tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
    res = {}
    for impl in ("hip", "torch", "torch64"):
        dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
        dec.load_state_dict(sd, strict=True)
        dec = dec.cuda().train()
        tp, tg = torch.from_numpy(z.copy()).cuda(), torch.from_numpy(g.copy()).cuda()
        if impl == "torch64":
            dec, tp, tg = dec.double(), tp.double(), tg.double()
        tp.requires_grad_(True); tg.requires_grad_(True)
        ps, mus, lvs = dec(tp, tg, mode=mode) if impl == "hip" else dec.forward_torch(tp, tg, mode=mode)
        pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
        loss = nets.PointFlowNLL()([tp] + ps, [pm] + mus, [pl] + lvs) + 0.1 * (ps[2] * mus[4]).mean()
        loss.backward()
        res[impl] = dict(gg=tg.grad, gp=tp.grad, grads={k: v.grad for k, v in dec.named_parameters() if v.grad is not None})
    h, t, t32 = res["hip"], res["torch64"], res["torch"]
    gg_h, gg_32 = rel(h["gg"], t["gg"]) * N, rel(t32["gg"], t["gg"]) * N
    ph = max(rel(h["grads"][k], t["grads"][k]) for k in t["grads"]) * B * N
    p32 = max(rel(t32["grads"][k], t["grads"][k]) for k in t["grads"]) * B * N
    rows.append((seed, gg_h, gg_32, ph, p32))
    print("seed %3d   d/dg error x N: hip %.2f  fp32 tensor ops %.2f      worst parameter gradient error x P: hip %.1f  fp32 tensor ops %.1f" % rows[-1], flush=True)
a = np.array(rows)
print("max over %d seeds:   d/dg x N: hip %.2f, fp32 tensor ops %.2f;   parameters x P: hip %.1f, fp32 tensor ops %.1f" % (len(a), a[:, 1].max(), a[:, 2].max(), a[:, 3].max(), a[:, 4].max()))
