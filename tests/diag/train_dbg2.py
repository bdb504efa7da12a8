"""debug: per-layer error growth of the HIP training path vs the tensor-op path (decoder golden config)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import flow_oracle as FO
from dpf_nets_amd import networks as nets
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
n_flows, B, N, G, seed = 2, 4, 96, 128, 11
if len(sys.argv) > 2: B, N = int(sys.argv[1]), int(sys.argv[2])
sd = FO.to_torch(FO.make_decoder_state(seed, n_flows, 64, G))
tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
res = {}
for impl in ("hip", "torch", "torch64"):
    dec = nets.LocalCondRNVPDecoder(n_flows, 64, G, weight_std=0.01)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().train()
    tp = torch.from_numpy(tgt.copy()).cuda(); tg = torch.from_numpy(g.copy()).cuda()
    if impl == "torch64":
        dec = dec.double(); tp = tp.double(); tg = tg.double()
    with torch.no_grad():
        res[impl] = dec(tp, tg, mode="inverse") if impl == "hip" else dec.forward_torch(tp, tg, mode="inverse")
    res[impl + "_sd"] = {k: v.clone() for k, v in dec.state_dict().items() if "running" in k}
for l in range(5, -1, -1):
    print("layer", l, " ".join("%s hip %.1e torch32 %.1e |" % (nm, rel(res["hip"][i][l], res["torch64"][i][l]),
                                                           rel(res["torch"][i][l], res["torch64"][i][l]))
                               for i, nm in enumerate(("ps", "mus", "lvs"))))
worst = sorted(((rel(res["hip_sd"][k], res["torch64_sd"][k]), rel(res["torch_sd"][k], res["torch64_sd"][k]), k) for k in res["torch_sd"]), reverse=True)[:6]
for w in worst: print("  stat %.1e (torch32 %.1e) %s" % w)
