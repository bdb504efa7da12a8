"""approx-EMD timing / roofline at the BASELINE sizes (cfg-2 and the cfg-5 stress shape)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import synthetic as SY                         # noqa: E402
from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK   # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    for (B, N, reps) in ((32, 2048, 3), (2, 8192, 3), (16, 8192, 1)):
        a = torch.from_numpy(SY.uniform_f32(1, (B, N, 3), -0.25, 0.25)).cuda()
        b = torch.from_numpy(SY.uniform_f32(2, (B, N, 3), -0.25, 0.25)).cuda()
        match, temp = BK.ApproxMatch(a, b)
        t_match = timed(lambda: BK.ApproxMatch(a, b), reps)
        t_cost = timed(lambda: BK.MatchCost(a, b, match), reps)
        t_fused = timed(lambda: BK.ApproxMatchCost(a, b), reps)
        t_grad = timed(lambda: BK.MatchCostGrad(a, b, match), reps)
        BK.EMD_GRAD_TWO_PASS = True
        t_grad2 = timed(lambda: BK.MatchCostGrad(a, b, match), reps)
        BK.EMD_GRAD_TWO_PASS = False
        nm = float(B) * N * N
        print("B=%d N=%d  approxmatch+cost fused %.2f ms |  approxmatch %.2f ms (%.0f GB/s of the 76*n*m RMW model, %.2e exp/s)  matchcost %.3f ms (%.0f GB/s)  "
              "grad one-pass %.3f ms (%.0f GB/s of 4*n*m) | two-pass %.3f ms" %
              (B, N, t_fused, t_match, 76 * nm / t_match / 1e6, 36 * nm / t_match * 1e3, t_cost, 4 * nm / t_cost / 1e6, t_grad,
               4 * nm / t_grad / 1e6, t_grad2), flush=True)


if __name__ == "__main__":
    main()
