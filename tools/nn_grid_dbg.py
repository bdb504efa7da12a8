import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
import bench
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
rng = np.random.default_rng(0)
B, N = 32, 2048
cases = {"uniform": (rng.random((B, N, 3), dtype=np.float32), rng.random((B, N, 3), dtype=np.float32)),
         "gauss": (rng.standard_normal((B, N, 3)).astype(np.float32), rng.standard_normal((B, N, 3)).astype(np.float32)),
         "gauss_vs_shifted": (rng.standard_normal((B, N, 3)).astype(np.float32), (rng.standard_normal((B, N, 3)) * 0.3 + 2).astype(np.float32))}
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
with torch.no_grad():
    dec.eval()
    ps, mus, lvs = dec(z, g, mode="direct", n_layers=args.layers)
out = ps[-1].transpose(1, 2).contiguous()
print("flow out: min", out.amin((0, 1)).tolist(), "max", out.amax((0, 1)).tolist(), "std", out.std((0, 1)).tolist())
print("target  : min", tgt_pm.amin((0, 1)).tolist(), "max", tgt_pm.amax((0, 1)).tolist(), "std", tgt_pm.std((0, 1)).tolist())
cases["bench"] = (out.cpu().numpy(), tgt_pm.cpu().numpy())
def surf(B, N):       # points on a torus surface with noise: a stand-in for mesh-sampled clouds
    u, v = rng.random((B, N)) * 2 * np.pi, rng.random((B, N)) * 2 * np.pi
    x = (0.35 + 0.12 * np.cos(v)) * np.cos(u); y = (0.35 + 0.12 * np.cos(v)) * np.sin(u); z = 0.12 * np.sin(v)
    return (np.stack([x, y, z], -1) + rng.standard_normal((B, N, 3)) * 0.004).astype(np.float32)
cases["surface"] = (surf(B, N), surf(B, N))
B2, N2 = 16, 8192
cases["uniform 16x8192"] = (rng.random((B2, N2, 3), dtype=np.float32), rng.random((B2, N2, 3), dtype=np.float32))
cases["gauss 16x8192"] = (rng.standard_normal((B2, N2, 3)).astype(np.float32), rng.standard_normal((B2, N2, 3)).astype(np.float32))
cases["surface 16x8192"] = (surf(B2, N2), surf(B2, N2))
for name, (a, b) in cases.items():
    ta, tb = torch.from_numpy(np.ascontiguousarray(a)).cuda(), torch.from_numpy(np.ascontiguousarray(b)).cuda()
    res = {}
    for impl in ("brute", "grid"):
        BK.NN_IMPL = impl
        res[impl] = t(lambda: BK.NNDistance(ta, tb))
        r = BK.NNDistance(ta, tb)
        if impl == "brute": ref = r
        else: assert all(torch.equal(x, y) for x, y in zip(r, ref)), name
    print("%-18s brute %7.1f us   grid %7.1f us" % (name, res["brute"], res["grid"]))
