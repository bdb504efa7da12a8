"""Diagnostic for test_training_small_w1_vs_float64: where the error of one parameter gradient sits."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd import networks as nets
from dpf_nets_amd.networks import train_engine
from oracle import flow_oracle as FO
train_engine.TRAIN_PRECISION = os.environ.get("PREC", "f16x3")
shrink = 2.0 ** -int(sys.argv[1]) if len(sys.argv) > 1 else 2.0 ** -6
name = sys.argv[2] if len(sys.argv) > 2 else "flows.0.nvp3.T_logvar_0.logvar_sd1.weight"
B, N, G, seed = 4, 512, 128, 37
sd = FO.to_torch(FO.make_decoder_state(seed, 2, 64, G))
for k in sd:
    if k.endswith("_sd1.weight"):
        sd[k] = sd[k] * shrink
tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
res = {}
for impl in ("hip", "torch", "torch64"):
    dec = nets.LocalCondRNVPDecoder(2, 64, G, weight_std=0.01)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().train()
    tp, tg = torch.from_numpy(tgt.copy()).cuda(), torch.from_numpy(g.copy()).cuda()
    if impl == "torch64":
        dec, tp, tg = dec.double(), tp.double(), tg.double()
    tp.requires_grad_(True); tg.requires_grad_(True)
    if impl != "hip":
        lay, br = name.split(".")[:3], name.split(".")[3].split("_")[1]
        m = getattr(getattr(dec.flows[int(lay[1])], lay[2]), "T_%s_1" % br)[0]
        m.register_forward_pre_hook(lambda mod, inp, impl=impl: res.__setitem__(impl + "/y", inp[0].detach().double().cpu().numpy().copy()))
    ps, mus, lvs = dec(tp, tg, mode="inverse") if impl == "hip" else dec.forward_torch(tp, tg, mode="inverse")
    pm, pl = torch.zeros(B, 3, N).cuda().to(tp.dtype), torch.full((B, 3, N), -3.6).cuda().to(tp.dtype)
    nets.PointFlowNLL()(ps + [tp], [pm] + mus, [pl] + lvs).backward()
    res[impl] = {k: v.grad.double().cpu().numpy() for k, v in dec.named_parameters() if v.grad is not None}
    res[impl + "/out"] = [x.detach().double().cpu().numpy() for x in ps + mus + lvs]
h, t32, t = res["hip"], res["torch"], res["torch64"]
for i, (a, b, c) in enumerate(zip(res["hip/out"], res["torch/out"], res["torch64/out"])):
    sc = np.abs(c).max() + 1e-30
    print("OUT %d hip %.2e fp32 %.2e" % (i, np.abs(a - c).max() / sc, np.abs(b - c).max() / sc))
e, e32 = np.abs(h[name] - t[name]), np.abs(t32[name] - t[name])
s = np.abs(t[name]).max()
print(name, "scale", s, "hip max err", e.max() / s, "fp32 max err", e32.max() / s)
print("rows (output channel) err/s:", np.round(e.max(axis=1) / s, 5)[:64])
print("row scales:", np.round(np.abs(t[name]).max(axis=1) / s, 3)[:64])
rows = []
for k in t:
    sc = np.abs(t[k]).max()
    if sc > 0:
        rows.append((np.abs(h[k] - t[k]).max() / sc, np.abs(t32[k] - t[k]).max() / sc, k))
rows.sort(reverse=True)
for r in rows[:int(os.environ.get("TOP", "14"))]:
    print("TOP %.2e %.2e %s" % r)
ee = (h[name] - t[name]).reshape(64, 64) / s
print("COL err:", np.round(np.abs(ee).max(axis=0), 5))
print("ROW err:", np.round(np.abs(ee).max(axis=1), 5))
r = np.abs(ee).max(axis=1)
c = int(r.argmax())
print("WORST ROW", c, r[c], "second", np.sort(r)[-2], "row true scale", np.abs(t[name].reshape(64, 64)[c]).max() / s)
print("ratio err/true in row:", np.round(ee[c] / (t[name].reshape(64, 64)[c] / s + 1e-12), 3)[:16])
for k in h:
    if k.startswith(name.rsplit(".", 2)[0][:-1]) or "nvp3.T_logvar_0" in k:
        v = (h[k] - t[k]).reshape(-1); tt = t[k].reshape(-1)
        if v.size in (64, 128):
            i = int(np.abs(v).argmax()); print("  ", k, "worst idx", i, v[i], tt[i])

y64, y32 = res["torch64/y"], res["torch/y"]
print("y shape", y64.shape)
for ch in (c, (c + 1) % 64):
    a = y64[:, ch, :].reshape(-1); b = y32[:, ch, :].reshape(-1)
    rms = np.sqrt((a * a).mean())
    o = np.argsort(np.abs(a))[:6]
    print("ch", ch, "rms", rms, "smallest |y|/rms:", np.abs(a[o]) / rms, "fp32 err there", (b[o] - a[o]) / rms, "fp32 sign flips", int(((a > 0) != (b > 0)).sum()),
          "max fp32 err/rms", np.abs(b - a).max() / rms)
