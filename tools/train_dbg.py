"""debug: per-case error of the HIP training path vs golden and vs the tensor-op path"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as FO
from oracle.gen_golden import layer_inputs
from dpf_nets_amd import networks as nets
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
gold = np.load(os.path.join(gd, "flow_layer.npz")); meta = json.load(open(os.path.join(gd, "flow_layer.json")))
B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
def rel(a, b):
    a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().cpu().numpy() if torch.is_tensor(b) else b
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
for case in meta["cases"]:
    if case["bn"] != "train": continue
    t = case["tag"]
    out = {}
    for impl in ("hip", "torch"):
        mod = nets.CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
        mod.load_state_dict(FO.to_torch(FO.make_layer_state(case["seed"], F, G, case["warp"])), strict=True)
        mod = mod.cuda().train()
        p, g, r1, r2, r3 = layer_inputs(case["seed"], B, N, G)
        tp = torch.from_numpy(p.copy()).cuda().requires_grad_(True)
        tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
        po, mu, lv = mod(tp, tg, mode=case["mode"]) if impl == "hip" else mod.forward_torch(tp, tg, mode=case["mode"])
        loss = (po * torch.from_numpy(r1).cuda()).sum() + (lv * torch.from_numpy(r2).cuda()).sum() + (mu * torch.from_numpy(r3).cuda()).sum()
        loss.backward()
        out[impl] = (po, tp.grad, tg.grad, {k: v.grad for k, v in mod.named_parameters()})
    h, tt = out["hip"], out["torch"]
    d = (h[1] - tt[1]).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print(t, "po %.1e/%.1e" % (rel(h[0], gold[t + "/p_out"]), rel(tt[0], gold[t + "/p_out"])),
          "gp %.1e/%.1e" % (rel(h[1], gold[t + "/grad_p"]), rel(tt[1], gold[t + "/grad_p"])),
          "gg %.1e/%.1e" % (rel(h[2], gold[t + "/grad_g"]), rel(tt[2], gold[t + "/grad_g"])),
          "argmax", idx, "n>1e-4:", int((d > 1e-4 * tt[1].abs().max()).sum()))
    worst = sorted(((rel(h[3][k], tt[3][k]), k) for k in tt[3]), reverse=True)[:3]
    print("    worst param grads:", ["%s %.1e" % (k, r) for r, k in worst])
