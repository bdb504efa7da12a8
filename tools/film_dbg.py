import sys, torch, numpy as np
sys.path.insert(0, '.')
from dpf_nets_amd import synthetic as SY
from dpf_nets_amd.networks import LocalCondRNVPDecoder
B, N, G, nf = 32, 2048, 128, 1
state = SY.make_decoder_state(5, nf, 64, G)
dec = LocalCondRNVPDecoder(nf, 64, G)
dec.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()})
dec = dec.cuda().eval()
tgt, z, g = SY.synthetic_inputs(5, B, N, G)
tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
with torch.no_grad():
    a = dec(tz, tg)[0].stacked.clone()
    b = dec(tz, tg)[0].stacked.clone()
    print("run-to-run equal:", torch.equal(a, b))
    for lo, hi in ((0, 8), (5, 9), (8, 16), (3, 4), (0, 32)):
        s = dec(tz[lo:hi].contiguous(), tg[lo:hi].contiguous())[0].stacked
        d = (s - a[:, lo:hi]).abs()
        print("sub", lo, hi, "equal:", torch.equal(s, a[:, lo:hi]), "maxdiff", float(d.max()), "per-cloud", [float(x) for x in d.amax(dim=(0, 2, 3))])
