"""Fused encoder kernel vs the tensor-op path on the same GPU (eval mode), at the BASELINE sizes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import networks as nets, synthetic as SY       # noqa: E402

FLOP_PER_POINT = 2 * (3 * 64 + 64 * 128 + 128 * 256 + 256 * 512)    # 344 KFLOP


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    torch.manual_seed(0)
    enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda().eval()
    for bn in (enc.features.init_sd_bn, enc.features.sd0_bn, enc.features.sd1_bn, enc.features.sd2_bn):
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32, 2048), (64, 2048), (16, 8192), (4, 256)]
    for (B, N) in shapes:
        x = torch.from_numpy(SY.uniform_f32(1, (B, 3, N), -0.25, 0.25)).cuda()
        with torch.no_grad():
            line = "B=%d N=%d:" % (B, N)
            for prec in ("bf16x3", "bf16x6", "bf16"):
                enc.precision = prec
                t = timed(lambda: torch.max(enc(x), dim=2)[0])
                line += "  %s %.1f us (%.0f TFLOP/s algorithmic)" % (prec, t, FLOP_PER_POINT * B * N / t / 1e6)
            enc.precision = "bf16x3"
            tf = timed(lambda: enc(x).tensor())
            tt = timed(lambda: torch.max(enc.forward_torch(x), dim=2)[0], reps=5)
            line += "  | with (B,512,N) output %.1f us | tensor-op path %.1f us" % (tf, tt)
        print(line, flush=True)


if __name__ == "__main__":
    main()
