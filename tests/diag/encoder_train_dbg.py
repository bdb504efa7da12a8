"""Per-parameter gradient error of the training-mode encoder (csrc/encoder_train.hip) and of the fp32 tensor-op path against
float64 tensor ops: python tests/diag/encoder_train_dbg.py B N"""
import copy, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from dpf_nets_amd import networks as nets
from oracle import encoder_oracle as EO, flow_oracle as FO, detrng

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))

B, N = (int(v) for v in sys.argv[1:3])
enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512])
enc.load_state_dict(FO.to_torch(EO.make_encoder_state(11)), strict=True)
enc = enc.cuda().train()
ref = copy.deepcopy(enc).double(); ref.hip_training = False
t32 = copy.deepcopy(enc); t32.hip_training = False
x = torch.from_numpy(EO.encoder_inputs(11, B, N)).cuda()
r = torch.from_numpy(detrng.normal_f32(detrng.key(11, "enc_r"), (B, 512))).cuda()
out = {}
for name, m, xx, rr in (("hip", enc, x, r), ("f64", ref, x.double(), r.double()), ("t32", t32, x, r)):
    pooled = torch.max(m(xx), dim=2)[0]
    (pooled * rr).sum().backward()
    out[name] = pooled
print("pooled hip", rel(out["hip"], out["f64"]), "t32", rel(out["t32"], out["f64"]))
for (k, p), (_, q), (_, t) in zip(enc.named_parameters(), ref.named_parameters(), t32.named_parameters()):
    print("%-32s hip %.3e  t32 %.3e   |g|max %.3e" % (k, rel(p.grad, q.grad), rel(t.grad, q.grad), float(q.grad.abs().max())))
for (k, v), (_, u) in zip(enc.state_dict().items(), ref.state_dict().items()):
    if "running" in k:
        print("%-40s %.3e" % (k, rel(v, u)))
