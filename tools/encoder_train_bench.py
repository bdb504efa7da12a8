"""Training-mode PointNet encoder, forward + backward of `torch.max(enc(x), 2)[0]`: the tensor-op path next to the HIP path
(csrc/encoder_train.hip) at cfg-2's shape.  usage: python tools/encoder_train_bench.py [B N iters]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dpf_nets_amd import networks  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    B, N, iters = (int(v) for v in (sys.argv[1:4] + ["32", "2048", "30"][len(sys.argv) - 1:]))
    hiponly = "hiponly" in sys.argv
    torch.manual_seed(0)
    enc = networks.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda().train()
    x = (torch.rand(B, 3, N, device="cuda") - 0.5) * 0.5
    r = torch.randn(B, 512, device="cuda")

    def step(hip):
        enc.hip_training = hip
        for p in enc.parameters():
            p.grad = None
        out = torch.max(enc(x), dim=2)[0]
        (out * r).sum().backward()

    def fwd(hip):
        enc.hip_training = hip
        with torch.no_grad():
            torch.max(enc(x), dim=2)[0]

    res = {}
    for name, hip in (("hip", True),) if hiponly else (("torch", False), ("hip", True)):
        res[name + "_fwd_ms"] = timeit(lambda: fwd(hip), iters)
        res[name + "_fwd_bwd_ms"] = timeit(lambda: step(hip), iters)
    for prec in () if hiponly else ("bf16x3",):
        enc.train_precision = prec
        res["hip_%s_fwd_ms" % prec] = timeit(lambda: fwd(True), iters)
        res["hip_%s_fwd_bwd_ms" % prec] = timeit(lambda: step(True), iters)
    print(res)


if __name__ == "__main__":
    main()
