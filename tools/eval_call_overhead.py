"""Host-side cost of the eval-mode module calls (no graph capture): decoder forward, encoder + max, nn_distance."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import networks as nets, synthetic as SY       # noqa: E402
from dpf_nets_amd.metrics.StructuralLosses.nn_distance import nn_distance   # noqa: E402


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6


def main():
    B, N, G = 32, 2048, 128
    tgt, z, g = SY.synthetic_inputs(3, B, N, G)
    tz, tg, tp = (torch.from_numpy(v).cuda() for v in (z, g, tgt))
    tpm = tp.transpose(1, 2).contiguous()
    with torch.no_grad():
        for n_flows in (5, 21):
            dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().eval()
            for lists in (True, False):
                dec.materialize_lists = lists
                h, t = timed(lambda: dec(tz, tg, mode="direct"))
                print("decoder n_flows=%d lists=%s: host issue %.0f us, wall %.0f us per call" % (n_flows, lists, h, t))
        enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda().eval()
        h, t = timed(lambda: torch.max(enc(tp), dim=2)[0])
        print("encoder + max: host issue %.0f us, wall %.0f us" % (h, t))
        h, t = timed(lambda: nn_distance(tpm, tpm))
        print("nn_distance: host issue %.0f us, wall %.0f us" % (h, t))


if __name__ == "__main__":
    main()
