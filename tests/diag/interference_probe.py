"""Diagnostic (r06): do the product kernels that keep PACKED fp32 VALU instructions (no MFMA of their own: csrc/Makefile's gate) still
return the same bits while ANOTHER stream runs a kernel that issues MFMAs with gaps on the same SIMDs?  (tools/ubench/pk_vs_mfma_waves2.hip:
a packed fp32 instruction can lose the low half of its result in lanes 48-63 while another wave of its SIMD issues MFMAs.)

    python tests/diag/interference_probe.py [reps [gaps...]]

For every op: the result without interference (twice: must agree), then `reps` results while tools/interfere's kernel runs on a side
stream with `gap` wait states between its MFMAs; prints how many results differ and in how many elements."""
import ctypes, os, subprocess, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from dpf_nets_amd._lib import lib
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
from oracle.gen_golden import chamfer_inputs

idir = os.path.join(root, "tools", "interfere")
so = os.path.join(idir, "libmfma_interferer.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(idir, "mfma_interferer.hip"), "-o", so], check=True)
I = ctypes.CDLL(so)
I.interferer_launch.restype = ctypes.c_int
I.interferer_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gaps = [int(v) for v in sys.argv[2:]] or [0, 4, 15]
side = torch.cuda.Stream()
stop = torch.zeros(1, dtype=torch.int32, device="cuda")
sink = torch.zeros(8192, device="cuda")


class Interference:
    def __init__(self, gap, blocks=2048):
        self.gap, self.blocks = gap, blocks

    def __enter__(self):
        stop.zero_()
        torch.cuda.synchronize()
        rc = I.interferer_launch(self.blocks, self.gap, stop.data_ptr(), 40000, sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc

    def __exit__(self, *a):
        stop.fill_(1)
        torch.cuda.synchronize()
        it = sink[:self.blocks]
        self.iters = (float(it.min()), float(it.max()))      # < 40000: stopped by the flag, i.e. alive to the end


def nn_op(impl, B, n, m, seed):
    a, b = chamfer_inputs(seed, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def run():
        old = BK.NN_IMPL
        BK.NN_IMPL = "brute" if impl == "small" else impl
        old_small = lib().dpf_nn_small_mode(1 if impl == "small" else -1)
        try:
            out = BK.NNDistance(ta, tb)
        finally:
            BK.NN_IMPL = old
            lib().dpf_nn_small_mode(old_small)
        return [o.clone() for o in out]
    return run


def emd_op(matrix, B, n, m, seed):
    a, b = chamfer_inputs(seed, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def run():
        prev = lib().dpf_emd_set_matrix_path(1 if matrix else 0)
        try:
            match, temp, cost = BK.ApproxMatchCost(ta, tb)
        finally:
            lib().dpf_emd_set_matrix_path(prev)
        return [match.clone(), cost.clone()]
    return run


def control_op(blocks=1024, iters=4000):
    """the positive control: packed chains that check themselves (tools/interfere/mfma_interferer.hip) -> [number of wrong results]"""
    I.pk_chain_check.restype = ctypes.c_int
    I.pk_chain_check.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")

    def run():
        bad.zero_()
        rc = I.pk_chain_check(blocks, iters, bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return [bad.clone()]
    return run


OPS = [("POSITIVE CONTROL: self-checking packed chains", control_op()),
       ("POSITIVE CONTROL, 32 workgroups x 20 chains", control_op(32, 20)),
       ("POSITIVE CONTROL, 256 workgroups x 200 chains", control_op(256, 200)),
       ("nndistance scan   B=8  2048x2048", nn_op("brute", 8, 2048, 2048, 11)),
       ("nndistance small  B=4  2048x2048", nn_op("small", 4, 2048, 2048, 12)),
       ("nndistance filter B=32 2048x2048", nn_op("mfma", 32, 2048, 2048, 13)),
       ("approx-EMD packed-VALU family B=4 1024x1024", emd_op(False, 4, 1024, 1024, 14)),
       ("approx-EMD matrix-core family B=4 1024x1024", emd_op(True, 4, 1024, 1024, 14))]


def same(x, y):
    return all(torch.equal(p.view(torch.int32) if p.dtype == torch.float32 else p, q.view(torch.int32) if q.dtype == torch.float32 else q)
               for p, q in zip(x, y))


def ndiff(x, y):
    return sum(int((p.view(torch.int32) != q.view(torch.int32)).sum()) if p.dtype == torch.float32 else int((p != q).sum()) for p, q in zip(x, y))


for name, op in OPS:
    if name.startswith("POSITIVE"):
        print("(positive control: 1024 x 256 lanes x 4000 chains of 4 packed fmas = 1.6e7 wave-level packed instructions per launch; `max` = wrong lane results in one launch)")
    try:
        ref = op()
    except Exception as e:      # noqa: BLE001
        print("%-48s skipped: %r" % (name, e))
        continue
    torch.cuda.synchronize()
    again = op()
    torch.cuda.synchronize()
    line = "%-48s alone: %s" % (name, "repeats" if same(ref, again) else "DIFFERS (%d elements)" % ndiff(ref, again))
    for g in gaps:
        bad, worst = 0, 0
        ctx = Interference(g)
        with ctx:
            for r in range(reps):
                out = op()
                torch.cuda.current_stream().synchronize()      # (THIS stream only: a device-wide wait would wait for the interferer)
                d = ndiff(ref, out)
                bad += d != 0
                worst = max(worst, d)
                if name.startswith("POSITIVE"):
                    worst = max(worst, int(out[0]))              # (the control's own count of wrong results in this launch)
        line += " | gap %2d: %d of %d differ (max %d elements; interferer ran %g-%g iterations)" % ((g, bad, reps, worst) + ctx.iters)
    print(line, flush=True)
