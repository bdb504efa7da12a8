"""Diagnostic: bit-stability of the MFMA kernels outside approx-EMD over many repeats at low and high occupancy (r05: the EMD
passes flickered in builds whose first-of-chain MFMA had a literal-zero C and its destination on a source's registers; the flow,
training, encoder and Chamfer kernels contain such instructions too -- tools/mfma_overlap_check.py).   repeat_all.py [reps]"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd import networks as nets
from dpf_nets_amd import synthetic as SY
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def same(a, b):
    return all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
               for x, y in zip(a, b))


for (B, N) in ((2, 40), (2, 300), (4, 2048), (8, 2048), (32, 2048)):
    torch.manual_seed(0)
    dec = nets.LocalCondRNVPDecoder(2, 64, 128).cuda().train(); dec.flatten_parameters()
    tgt, z, g = SY.synthetic_inputs(7, B, N, 128)
    tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    first, bad = None, 0
    for it in range(reps):
        dec.zero_grad(set_to_none=True)
        tpi = tp.clone().requires_grad_(True)
        ps, mus, lvs = dec(tpi, tg, mode="inverse")
        (ps[0].square().mean() + sum(lvs).mean()).backward()
        got = [ps[0].detach().clone(), tpi.grad.clone()] + [p.grad.clone() for p in dec.parameters() if p.grad is not None]
        if first is None: first = got
        elif not same(got, first): bad += 1
    print("training step  B=%2d N=%4d: %d of %d repeats differ" % (B, N, bad, reps - 1))
    dec.eval()
    first, bad = None, 0
    with torch.no_grad():
        for it in range(reps):
            ps, mus, lvs = dec(tp, tg, mode="direct")
            got = [ps[-1].clone(), lvs[-1].clone()]
            if first is None: first = got
            elif not same(got, first): bad += 1
    print("eval stack     B=%2d N=%4d: %d of %d repeats differ" % (B, N, bad, reps - 1))
    a = tp.transpose(1, 2).contiguous(); b = (a.flip(1) + 0.01).contiguous()
    first, bad = None, 0
    for it in range(reps):
        got = [t.clone() for t in BK.NNDistance(a, b)]
        if first is None: first = got
        elif not same(got, first): bad += 1
    print("nn_distance    B=%2d N=%4d: %d of %d repeats differ" % (B, N, bad, reps - 1))
for (B, N) in ((2, 300), (8, 2048), (32, 2048)):
    torch.manual_seed(1)
    enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda()
    if enc is None: break
    x = torch.randn(B, 3, N, device="cuda")
    for mode in ("eval", "train"):
        enc.train(mode == "train")
        first, bad = None, 0
        for it in range(reps):
            enc.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(mode == "train")
            out = enc(xi)
            out = out if torch.is_tensor(out) else out[0]
            got = [out.detach().clone()]
            if mode == "train":
                out.square().mean().backward()
                got += [p.grad.clone() for p in enc.parameters() if p.grad is not None]
            if first is None: first = got
            elif not same(got, first): bad += 1
        print("encoder %-5s  B=%2d N=%4d: %d of %d repeats differ" % (mode, B, N, bad, reps - 1))
