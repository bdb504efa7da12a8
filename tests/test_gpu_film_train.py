"""The fused training-mode FiLM conditioner nets (csrc/film_train.hip) against the batched tensor ops they replace
(lib/networks/flows.py:33-45, 68-80 and autograd), through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F = 64


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd._lib import lib, check, current_stream
    return lib(), check, current_stream


def _reference(g, W0, gam, bet, W1, b1, eps, dfm):
    g = g.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (W0, gam, bet, W1, b1)]
    W0r, gamr, betr, W1r, b1r = ps
    u = torch.matmul(g.unsqueeze(0), W0r.transpose(1, 2))
    var, mean = torch.var_mean(u, dim=1, unbiased=False, keepdim=True)
    xhat = (u - mean) * torch.rsqrt(var + eps)
    y = xhat * gamr.unsqueeze(1) + betr.unsqueeze(1)
    fm = torch.baddbmm(b1r.unsqueeze(1), y * torch.sigmoid(y), W1r.transpose(1, 2))
    (fm * dfm).sum().backward()
    B = g.shape[0]
    return dict(fm=fm.detach(), xhat=xhat.detach(), mean=mean.detach().squeeze(1), uvar=var.detach().squeeze(1) * (B / (B - 1.0)),
                dg=g.grad, dW0=W0r.grad, dgam=gamr.grad, dbet=betr.grad, dW1=W1r.grad, db1=b1r.grad)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("K,B,G", [(12, 32, 128), (8, 5, 128), (4, 64, 512), (6, 17, 36), (252, 32, 128), (3, 2, 128)])
def test_fused_film_nets_vs_tensor_ops(K, B, G):
    L_, check, current_stream = _gpu()
    gen = torch.Generator().manual_seed(K * 1000 + B * 10 + G)
    dev = "cuda"
    g = torch.randn(B, G, generator=gen).to(dev)
    W0 = (torch.randn(K, F, G, generator=gen) / G ** 0.5).to(dev)
    gam = (1.0 + 0.2 * torch.randn(K, F, generator=gen)).to(dev)
    bet = (0.3 * torch.randn(K, F, generator=gen)).to(dev)
    W1 = (torch.randn(K, F, F, generator=gen) / 8.0).to(dev)
    b1 = (0.1 * torch.randn(K, F, generator=gen)).to(dev)
    dfm = torch.randn(K, B, F, generator=gen).to(dev)
    eps = 1e-5
    ref = _reference(g, W0, gam, bet, W1, b1, eps, dfm)
    fm, xhat = torch.empty(K, B, F, device=dev), torch.empty(K, B, F, device=dev)
    rstd, mean, uvar = (torch.empty(K, F, device=dev) for _ in range(3))
    check(L_.dpf_film_train_forward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(),
                                    b1.data_ptr(), eps, fm.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), mean.data_ptr(),
                                    uvar.data_ptr(), current_stream()), "film_train_forward")
    tol = 2e-5 if B > 2 else 2e-3                 # B = 2: xhat = +-1 up to rstd, which amplifies the rounding of a tiny variance
    assert rel(fm, ref["fm"]) <= tol and rel(xhat, ref["xhat"]) <= tol
    assert rel(mean, ref["mean"]) <= 1e-5 and rel(uvar, ref["uvar"]) <= 1e-5
    outs = {}
    for accumulate in (0, 1):
        dW0, dW1 = torch.full_like(W0, 0.5), torch.full_like(W1, 0.5)
        dgam, dbet, db1 = (torch.full((K, F), 0.5, device=dev) for _ in range(3))
        dgp = torch.empty(K, B, G, device=dev)
        check(L_.dpf_film_train_backward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(),
                                         xhat.data_ptr(), rstd.data_ptr(), dfm.data_ptr(), dW0.data_ptr(), dgam.data_ptr(),
                                         dbet.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dgp.data_ptr(), accumulate,
                                         current_stream()), "film_train_backward")
        outs[accumulate] = dict(dW0=dW0, dgam=dgam, dbet=dbet, dW1=dW1, db1=db1, dg=dgp.sum(0))
    btol = 5e-5 if B > 2 else 5e-3
    for k in ("dW0", "dgam", "dbet", "dW1", "db1", "dg"):
        assert rel(outs[0][k], ref[k]) <= btol, (k, rel(outs[0][k], ref[k]))
        if k != "dg":                             # accumulate = 1 adds to what was there (0.5 everywhere)
            assert torch.allclose(outs[1][k] - 0.5, outs[0][k], rtol=0, atol=2e-6 * float(ref[k].abs().max()) + 1e-7), k


def test_fused_film_nets_refuse_what_they_are_not_built_for():
    L_, check, current_stream = _gpu()
    assert L_.dpf_film_train_max_batch() == 64
    t = torch.zeros(16, device="cuda")
    p = t.data_ptr()
    assert L_.dpf_film_train_forward(1, 65, 128, p, p, p, p, p, p, 1e-5, p, p, p, p, p, current_stream()) == -2      # DPF_ENOSUP
    assert L_.dpf_film_train_forward(1, 8, 126, p, p, p, p, p, p, 1e-5, p, p, p, p, p, current_stream()) == -2
    assert L_.dpf_film_train_forward(1, 1, 128, p, p, p, p, p, p, 1e-5, p, p, p, p, p, current_stream()) == -1       # DPF_EINVAL


def test_fused_running_statistics_update_has_the_tensor_ops_bits():
    """dpf_flow_train_update_running against `rm.mul_(1 - m); rm.add_(batch, alpha=m)` (FlatStore.update_running) bit for bit."""
    L_, check, current_stream = _gpu()
    L, m = 5, 0.1
    st_floats = int(L_.dpf_flow_train_stats_floats())
    gen = torch.Generator().manual_seed(5)
    dev = "cuda"
    film_mean, film_uvar = torch.randn(4 * L, F, generator=gen).to(dev), torch.rand(4 * L, F, generator=gen).to(dev)
    stats = torch.randn(L, st_floats, generator=gen).to(dev)
    rm0, rv0 = torch.randn(8 * L, F, generator=gen).to(dev), (0.5 + torch.rand(8 * L, F, generator=gen)).to(dev)
    nbt0 = torch.arange(8 * L, dtype=torch.int64, device=dev)
    # the tensor-op form
    rm, rv, nbt = rm0.clone(), rv0.clone(), nbt0.clone()
    sv = stats[:, :2 * 6 * F].view(L, 2, 6, F)
    flow_mean, flow_uvar = sv[:, :, (0, 2)].reshape(4 * L, F), sv[:, :, (4, 5)].reshape(4 * L, F)
    rm.mul_(1.0 - m); rv.mul_(1.0 - m)
    rm[:4 * L].add_(film_mean, alpha=m); rm[4 * L:].add_(flow_mean, alpha=m)
    rv[:4 * L].add_(film_uvar, alpha=m); rv[4 * L:].add_(flow_uvar, alpha=m)
    nbt.add_(1)
    # the kernel
    rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), nbt0.clone()
    check(L_.dpf_flow_train_update_running(L, m, film_mean.data_ptr(), film_uvar.data_ptr(), stats.data_ptr(), rm2.data_ptr(),
                                           rv2.data_ptr(), nbt2.data_ptr(), current_stream()), "update_running")
    assert torch.equal(rm, rm2) and torch.equal(rv, rv2) and torch.equal(nbt, nbt2)
