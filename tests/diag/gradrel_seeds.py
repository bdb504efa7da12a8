"""How far the HIP training stack's parameter gradients sit from float64, as multiples of the fp32 tensor ops' own distance, over
several seeds -- the body of tests/test_gpu_flow_train.py::test_training_hip_vs_tensor_op_path at (8, 2048, direct).
usage: gradrel_seeds.py <so-name> [seed ...]      (prints, per seed, the parameters above 2 x the fp32 path and above 4e-4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib
so = sys.argv[1]
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", so)
import numpy as np, torch
from oracle import flow_oracle as FO
from dpf_nets_amd import networks as nets

def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().cpu().numpy().astype(np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

B, N, mode, n_flows, G = 8, 2048, "direct", 2, 128
for seed in [int(x) for x in sys.argv[2:]] or [31, 32, 33, 34]:
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
        res[impl] = {k: v.grad for k, v in dec.named_parameters()}
    worst = []
    for k, v in res["torch64"].items():
        if v is None: continue
        r, r32 = rel(res["hip"][k], v), rel(res["torch"][k], v)
        if r > 4e-4 and r > 2 * r32: worst.append((r / max(r32, 1e-30), r, r32, k))
    if os.environ.get("SD2"):
        rr = []
        for k, v in res["torch64"].items():
            if v is None or not k.endswith("sd2.bias"): continue
            rr.append("%.1f" % (rel(res["hip"][k], v) / max(rel(res["torch"][k], v), 1e-30)))
        print(so, "seed", seed, "sd2.bias r/r32:", " ".join(rr))
        continue
    worst.sort(reverse=True)
    print(so, "seed", seed, "| " + "; ".join("%s %.1fx (%.1e / %.1e)" % (k.split("flows.")[1], f, r, r32) for f, r, r32, k in worst[:4]))
