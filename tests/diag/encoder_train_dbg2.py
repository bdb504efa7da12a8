"""Intermediates of the training-mode encoder kernels (y_l, dz_l in the workspace) against float64 autograd."""
import copy, ctypes, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from dpf_nets_amd import networks as nets
from dpf_nets_amd._lib import lib, check, current_stream
from oracle import encoder_oracle as EO, flow_oracle as FO, detrng

def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

B, N = (int(v) for v in sys.argv[1:3])
enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512])
enc.load_state_dict(FO.to_torch(EO.make_encoder_state(11)), strict=True)
enc = enc.cuda().train()
ref = copy.deepcopy(enc).double()
x = torch.from_numpy(EO.encoder_inputs(11, B, N)).cuda()
r = torch.from_numpy(detrng.normal_f32(detrng.key(11, "enc_r"), (B, 512))).cuda()
h = x.double(); ys, zs = [], []
for name in ("init_sd", "sd0", "sd1", "sd2"):
    y = getattr(ref.features, name)(h); y.retain_grad(); ys.append(y)
    z = getattr(ref.features, name + "_bn")(y); z.retain_grad(); zs.append(z)
    h = torch.relu(z)
pooled_ref = h.max(dim=2)[0]
(pooled_ref * r.double()).sum().backward()

L = lib()
canon = torch.cat([t.detach().reshape(-1) for t in enc._layer_tensors()]).contiguous()
ws = torch.zeros(L.dpf_encoder_train_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
pooled = torch.empty(B, 512, device="cuda")
check(L.dpf_encoder_train_forward(B, N, int(sys.argv[3]) if len(sys.argv) > 3 else 2, canon.data_ptr(), x.data_ptr(), ws.data_ptr(), pooled.data_ptr(), None, None, 0.1, current_stream()), "f")
dcanon = torch.zeros_like(canon)
check(L.dpf_encoder_train_backward(B, N, canon.data_ptr(), x.data_ptr(), ws.data_ptr(), pooled.data_ptr(), r.data_ptr(), dcanon.data_ptr(), current_stream()), "b")
torch.cuda.synchronize()
Np = (N + 31) // 32 * 32
C = [64, 128, 256, 512]
off, views = 0, {}
for nm, ls in (("y", range(4)), ("dz", range(3))):
    for l in ls:
        n = C[l] * B * Np
        views[nm + str(l)] = ws[off:off + 4 * n].view(torch.float32).view(B, C[l], Np)[:, :, :N]
        off += (4 * n + 255) // 256 * 256
print("pooled", rel(pooled, pooled_ref))
for l in range(4):
    print("y%d" % l, rel(views["y%d" % l], ys[l]))
for l in (2, 1, 0):
    d = views["dz%d" % l]
    print("dz%d" % l, rel(d, zs[l].grad), " sum-over-points err", rel(d.sum((0, 2)), zs[l].grad.sum((0, 2))),
          " |dz| l1 per feature", float(zs[l].grad.abs().sum((0, 2)).mean()), "dy ref", rel(0 * d, ys[l].grad) )
for l in (2, 1, 0):
    d, g = views["dz%d" % l].double(), zs[l].grad
    gm = g * (zs[l] > 0)          # reference dz (gradient w.r.t. the ReLU input) 
    flips = ((d != 0) != (gm != 0))
    agree = ~flips
    print("dz%d: mask disagreements %d of %d; error where they agree %.3e; |z| at disagreements (in std units) %s" % (
        l, int(flips.sum()), flips.numel(), float(((d - gm).abs() * agree).max() / gm.abs().max()),
        (zs[l][flips].abs() ).tolist()[:8]))
