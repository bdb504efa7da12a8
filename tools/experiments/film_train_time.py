"""Duration of the fused FiLM-net kernels against K, B, G (HIP events around 50 back-to-back launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dpf_nets_amd._lib import lib, check, current_stream
L_ = lib()
dev = "cuda"
F = 64
def run(K, B, G):
    g = torch.randn(B, G, device=dev); W0 = torch.randn(K, F, G, device=dev) / G ** 0.5
    gam, bet, b1 = torch.ones(K, F, device=dev), torch.zeros(K, F, device=dev), torch.zeros(K, F, device=dev)
    W1 = torch.randn(K, F, F, device=dev) / 8
    fm, xhat = torch.empty(K, B, F, device=dev), torch.empty(K, B, F, device=dev)
    rstd, mean, uvar = (torch.empty(K, F, device=dev) for _ in range(3))
    dfm = torch.randn(K, B, F, device=dev)
    dW0, dW1 = torch.zeros_like(W0), torch.zeros_like(W1)
    dgam, dbet, db1 = (torch.zeros(K, F, device=dev) for _ in range(3))
    def fwd():
        check(L_.dpf_film_train_forward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(), b1.data_ptr(),
                                        1e-5, fm.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), mean.data_ptr(), uvar.data_ptr(), current_stream()), "f")
    def bwd():
        check(L_.dpf_film_train_backward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(), xhat.data_ptr(),
                                         rstd.data_ptr(), dfm.data_ptr(), dW0.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), dW1.data_ptr(),
                                         db1.data_ptr(), None, 1, current_stream()), "b")
    out = []
    for fn in (fwd, bwd):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); e.synchronize()
        out.append(s.elapsed_time(e) / 50 * 1e3)
    print("K=%3d B=%2d G=%3d: forward %6.1f us  backward %6.1f us" % (K, B, G, out[0], out[1]))
for K, B, G in [(252, 32, 128), (252, 8, 128), (252, 64, 128), (252, 32, 512), (64, 32, 128), (8, 32, 128), (504, 32, 128)]:
    run(K, B, G)
