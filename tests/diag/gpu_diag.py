"""First-contact diagnostics on the GPU box: prints error statistics of every HIP kernel
against the CPU oracle / golden fixtures, and rough timings.  Not a test."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import flow_oracle as FO            # noqa: E402
from oracle import structural as S              # noqa: E402
from oracle.gen_golden import layer_inputs, chamfer_inputs   # noqa: E402
from dpf_nets_amd.networks import CondRealNVPFlow3D, LocalCondRNVPDecoder   # noqa: E402
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK   # noqa: E402


def err(got, ref):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    return "maxabs %.3e  rel-to-max %.3e  maxrel(|ref|>1e-3) %.3e" % (
        d.max(), d.max() / (np.abs(ref).max() + 1e-30),
        (d / np.maximum(np.abs(ref), 1e-3)).max())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    print(torch.cuda.get_device_name(0))
    gold = np.load(os.path.join(ROOT, "tests/golden/flow_layer.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests/golden/flow_layer.json")))
    B, N, F, G = meta["B"], meta["N"], meta["F"], meta["G"]
    for prec in ("bf16x6", "bf16x3", "bf16"):
        print("== single layer, precision", prec)
        for case in meta["cases"]:
            if case["bn"] != "eval":
                continue
            st = FO.make_layer_state(case["seed"], F, G, case["warp"])
            mod = CondRealNVPFlow3D(F, G, warp_inds=case["warp"])
            mod.load_state_dict(FO.to_torch(st), strict=True)
            mod = mod.cuda().eval()
            mod.precision = prec
            p, g, _, _, _ = layer_inputs(case["seed"], B, N, G)
            with torch.no_grad():
                po, mu, lv = mod(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda(), mode=case["mode"])
            t = case["tag"]
            print(" %-22s p_out %s" % (t, err(po, gold[t + "/p_out"])))
            print(" %-22s mu    %s" % ("", err(mu, gold[t + "/mu"])))
            print(" %-22s lv    %s" % ("", err(lv, gold[t + "/logvar"])))
    gold = np.load(os.path.join(ROOT, "tests/golden/flow_decoder.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests/golden/flow_decoder.json")))
    for prec in ("bf16x6", "bf16x3", "bf16"):
        print("== decoder, precision", prec)
        for case in meta["cases"]:
            if case.get("bn") == "train":
                continue
            c, nf, B, N, G, seed, mode = (case[k] for k in ("tag", "n_flows", "B", "N", "G", "seed", "mode"))
            dec = LocalCondRNVPDecoder(nf, 64, G)
            dec.load_state_dict(FO.to_torch(FO.make_decoder_state(seed, nf, 64, G)), strict=True)
            dec = dec.cuda().eval()
            dec.precision = prec
            tgt, z, g = FO.synthetic_inputs(seed, B, N, G)
            src = z if mode == "direct" else tgt
            with torch.no_grad():
                ps, mus, lvs = dec(torch.from_numpy(src).cuda(), torch.from_numpy(g).cuda(), mode=mode)
            for k in case["picks"]:
                print(" %-18s ps%-2d %s" % (c, k, err(ps[k], gold["%s/ps%d" % (c, k)])))
            print(" %-18s sumlv %s" % (c, err(lvs.total(), gold[c + "/sum_logvars"])))
    print("== chamfer")
    for (B, n, m) in ((3, 257, 257), (32, 2048, 2048)):
        a, b = chamfer_inputs(5, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d1, i1, d2, i2 = BK.NNDistance(ta, tb)
        r = S.nndistance(a, b)
        print(" (%d,%d,%d) dist1 exact %s idx1 exact %s dist2 exact %s idx2 exact %s  mismatches %d %d" % (
            B, n, m, np.array_equal(d1.cpu().numpy(), r[0]), np.array_equal(i1.cpu().numpy(), r[1]),
            np.array_equal(d2.cpu().numpy(), r[2]), np.array_equal(i2.cpu().numpy(), r[3]),
            int((i1.cpu().numpy() != r[1]).sum()), int((d1.cpu().numpy() != r[0]).sum())))
        print("   time %.1f us" % timeit(lambda: BK.NNDistance(ta, tb)))
    print("== emd")
    for (B, n, m) in ((2, 64, 64), (2, 128, 64), (3, 300, 300)):
        a, b = chamfer_inputs(6, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        match, _ = BK.ApproxMatch(ta, tb)
        cost = BK.MatchCost(ta, tb, match)
        rm, _ = S.approxmatch(a, b)
        print(" (%d,%d,%d) match %s" % (B, n, m, err(match, rm)))
        print("            cost  %s" % err(cost, S.matchcost(a, b, rm)))
    a, b = chamfer_inputs(6, 32, 2048, 2048)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    print("   emd fwd B=32 N=2048: %.1f us" % timeit(lambda: BK.MatchCost(ta, tb, BK.ApproxMatch(ta, tb)[0]), 3))
    print("== flow timing B=32 N=2048")
    for nf, L in ((5, 14), (5, 15), (21, 63)):
        dec = LocalCondRNVPDecoder(nf, 64, 128).cuda().eval()
        tgt, z, g = FO.synthetic_inputs(0, 32, 2048, 128)
        tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
        for prec in ("bf16", "bf16x3", "bf16x6"):
            dec.precision = prec
            for lists in (True, False):
                dec.materialize_lists = lists
                with torch.no_grad():
                    us = timeit(lambda: dec(tz, tg, mode="direct", n_layers=L))
                print(" L=%d %-7s lists=%-5s %.1f us  -> %.3e pts/s" % (L, prec, lists, us, 32 * 2048 / us * 1e6))


if __name__ == "__main__":
    main()
