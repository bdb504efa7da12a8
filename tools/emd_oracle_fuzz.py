"""The in-suite approx-EMD fuzz (tests/test_gpu_emd.py::emd_fuzz_case: nine cloud kinds, ragged shapes) WITHOUT its time box: N cases of
the default (matrix-core) path and every fourth also of the packed-VALU kernels against the CPU oracle -- and, per cloud, the SAME
auction in float64 (numpy), which tells how well the input is conditioned: the fp32 oracle's own distance from exact arithmetic.
The auction divides by (1e-9 + a sum of weights); a point whose neighbours have all been consumed has a sum of that size, and fp32
cannot resolve `remain - consumed` of a size-1 quantity to 1e-9: on such inputs EVERY fp32 evaluation -- the oracle, the reference's
__expf loop, both kernel families here -- returns its own rounding noise amplified, and a cost comparison measures that noise, not parity.
Per (kind, kernel family): clouds, worst cost error vs the oracle over ALL clouds, over the WELL-CONDITIONED ones (oracle within 1e-6
of the float64 auction) and the number of ill-conditioned clouds (oracle further than 1e-5 from float64) with the oracle's own worst
distance from float64 there.    python tools/emd_oracle_fuzz.py [N] [seed]"""
import os, sys, time
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tests.test_gpu_emd import emd_fuzz_case
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle import structural as S


def auction64(a1, b1):
    """approxmatch.cu:3-182 + matchcost in float64, whole passes as matrix expressions (tests/test_oracle_golden.py holds the C
    oracle to this reading)"""
    n, m = len(a1), len(b1)
    d2 = ((b1[:, None, :].astype(np.float64) - a1[None, :, :].astype(np.float64)) ** 2).sum(2)
    remL = np.full(n, 1.0 if n >= m else float(m // n)); remR = np.full(m, float(n // m) if n >= m else 1.0); match = np.zeros((m, n))
    for j in range(7, -2, -1):
        e = np.exp(-(4.0 ** j) * d2); ratioL = remL / (1e-9 + remR @ e); sumr = (e @ ratioL) * remR
        ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR; remR = np.maximum(0.0, remR - sumr)
        w = e * ratioR[:, None] * ratioL[None, :]; match += w; remL = np.maximum(0.0, remL - w.sum(0))
    return float((match * np.sqrt(d2)).sum())


N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
stat, t0 = {}, time.time()
for it in range(N):
    a, b, kind = emd_fuzz_case(rng)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    rm, _ = S.approxmatch(a, b); rc = S.matchcost(a, b, rm)
    r64 = np.array([auction64(a[i], b[i]) for i in range(len(a))])
    cond = np.abs(rc - r64) / np.maximum(np.abs(r64), 1e-6)                 # the oracle's own distance from exact arithmetic
    for fam in ((1, 0) if it % 4 == 0 else (1,)):
        lib().dpf_emd_set_matrix_path(fam)
        m, _, c = BK.ApproxMatchCost(ta, tb)
        ce = np.abs(c.cpu().numpy() - rc) / np.maximum(np.abs(rc), 1e-6)
        st = stat.setdefault((kind, "matrix" if fam else "valu"), dict(clouds=0, all=0.0, well=0.0, nwell=0, ill=0, illcond=0.0, illerr=0.0))
        st["clouds"] += len(ce); st["all"] = max(st["all"], float(ce.max()))
        well, ill = cond <= 1e-6, cond > 1e-5
        st["nwell"] += int(well.sum())
        if well.any(): st["well"] = max(st["well"], float(ce[well].max()))
        if ill.any():
            st["ill"] += int(ill.sum()); st["illcond"] = max(st["illcond"], float(cond[ill].max())); st["illerr"] = max(st["illerr"], float(ce[ill].max()))
    lib().dpf_emd_set_matrix_path(1)
print("# tools/emd_oracle_fuzz.py %d cases, %.0f s" % (N, time.time() - t0))
print("# kind       family clouds | worst cost error vs oracle: all clouds | well-conditioned clouds (count) | ill-conditioned: count, oracle's own worst distance from float64, worst error there")
for key in sorted(stat):
    st = stat[key]
    print("%-10s %-6s %5d | %.2e | %.2e (%d) | %d, %.2e, %.2e" % (key[0], key[1], st["clouds"], st["all"], st["well"], st["nwell"], st["ill"], st["illcond"], st["illerr"]))
