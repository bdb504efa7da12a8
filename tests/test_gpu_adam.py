"""dpf_adam_step (csrc/adam.hip): the fused AMSGrad-Adam step of SURVEY 8(f) rank 4 against the reference's op sequence
(lib/networks/optimizers.py:52-74 as networks/optimizers.py::Adam._update restates it, itself pinned bit for bit on the
reference's class by tests/golden/optimizer.npz): bit-identical over several steps, with and without AMSGrad / weight decay,
on sizes with and without a vector tail; and through the optimizer class on a flattened decoder."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


@pytest.mark.parametrize("n", [1 << 20, 4099, 3])
@pytest.mark.parametrize("amsgrad", [True, False])
@pytest.mark.parametrize("wd", [1e-6, 0.0])
def test_fused_adam_step_is_bitwise_the_op_sequence(n, amsgrad, wd):
    nets = _gpu()
    from dpf_nets_amd._lib import lib, current_stream
    gen = torch.Generator(device="cuda").manual_seed(n)
    nal = (n + 3) // 4 * 4                                              # 16-byte aligned views of bigger buffers
    p = torch.randn(nal, device="cuda", generator=gen)[:n]
    m = (torch.randn(nal, device="cuda", generator=gen) * 1e-2)[:n]
    v = (torch.rand(nal, device="cuda", generator=gen) * 1e-3)[:n]
    vm = (v * (1.0 + torch.rand(n, device="cuda", generator=gen))).contiguous()
    ref = [t.clone() for t in (p, m, v, vm)]
    got = [t.clone() for t in (p, m, v, vm)]
    lr, b1, b2, eps = 2.56e-4, 0.9, 0.999, 1e-8
    for step in range(1, 6):
        g = torch.randn(n, device="cuda", generator=gen) * (0.3 ** step)
        nets.Adam._update([ref[0]], [g], [ref[1]], [ref[2]], [ref[3]] if amsgrad else None, step, lr, b1, b2, eps, wd, amsgrad)
        rc = lib().dpf_adam_step(n, got[0].data_ptr(), g.data_ptr(), got[1].data_ptr(), got[2].data_ptr(),
                                 got[3].data_ptr() if amsgrad else None, lr, b1, b2, eps, wd, 1 - b1 ** step, math.sqrt(1 - b2 ** step),
                                 current_stream())
        assert rc == 0
        for name, a, b in zip(("p", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"), ref[:4 if amsgrad else 3], got):
            assert torch.equal(a, b), (step, name, int((a != b).sum()))
    if not amsgrad:
        assert torch.equal(got[3], vm)                                     # untouched without AMSGrad
    assert lib().dpf_adam_step(4, p.data_ptr() + 4, p.data_ptr(), p.data_ptr(), p.data_ptr(), None, lr, b1, b2, eps, wd, 0.1, 0.1, None) == -1   # misaligned
    assert lib().dpf_adam_step(4, p.data_ptr(), p.data_ptr(), p.data_ptr(), p.data_ptr(), None, lr, b1, b2, eps, wd, 0.0, 0.1, None) == -1       # bias correction 0


def test_fused_adam_step_on_the_reference_goldens_own_inputs(golden_dir):
    """the kernel one hop from the reference.  tests/golden/optimizer.npz holds what the REFERENCE's Adam
    (lib/networks/optimizers.py:15-76, run by oracle/gen_golden.py) leaves after five steps of the cyclic schedule on four
    parameters (a SharedDot-shaped weight, a vector, a matrix, a scalar) for every AMSGrad / weight-decay case; here dpf_adam_step
    itself takes the same parameters, gradients, learning rates and betas.  The golden was computed by ATen on the CPU, the kernel
    is bitwise the op sequence as ATen runs it on the GPU (test above): the two differ where the CPU's vectorised kernels contract a
    multiply-add -- measured after five steps: weights and moments within 2 ulp, most entries 0.  Bar: 4 ulp, at most half of the
    entries off at all."""
    import numpy as np
    import os
    _gpu()
    from dpf_nets_amd._lib import lib, current_stream
    from oracle import optimizer_oracle as OO
    from oracle.gen_golden import OPT_CASES, optimizer_inputs
    gold = np.load(os.path.join(golden_dir, "optimizer.npz"))
    for name, (ams, wd) in OPT_CASES.items():
        shapes = [np.asarray(v).shape for v in optimizer_inputs(5, -1)]
        ps = [torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(-1).cuda() for v in optimizer_inputs(5, -1)]
        # (every parameter in its own 16-byte aligned buffer: the flat store of the training engine aligns them the same way)
        ms = [torch.zeros_like(p) for p in ps]
        vs = [torch.zeros_like(p) for p in ps]
        vmax = [torch.zeros_like(p) for p in ps]
        for step in range(5):
            lr, betas = OO.lr_update(4, 3, 1e-4, 2e-3, 0.9, 0.99, 0.999, step // 4, step % 4)
            np.testing.assert_allclose([lr, betas[1]], gold[name + "_sched"][step], rtol=1e-15)
            gs = [torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)).reshape(-1).cuda() for g in optimizer_inputs(5, step)]
            for p, g, m, v, vm in zip(ps, gs, ms, vs, vmax):
                rc = lib().dpf_adam_step(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), vm.data_ptr() if ams else None,
                                         lr, betas[0], betas[1], 1e-8, wd, 1 - betas[0] ** (step + 1), math.sqrt(1 - betas[1] ** (step + 1)),
                                         current_stream())
                assert rc == 0
        torch.cuda.synchronize()
        for i, shp in enumerate(shapes):
            for what, got in (("p", ps[i]), ("m", ms[i]), ("v", vs[i])) + ((("vmax", vmax[i]),) if ams else ()):
                ref = np.asarray(gold["%s_%s%d" % (name, what, i)], dtype=np.float32).reshape(-1)
                ulp = np.abs(got.cpu().numpy().view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
                assert int(ulp.max()) <= 4, (name, what, i, int(ulp.max()))
                assert ulp.size < 8 or float((ulp > 0).mean()) <= 0.5, (name, what, i, float((ulp > 0).mean()))


def test_optimizer_class_takes_the_fused_path_and_keeps_the_trajectory(monkeypatch):
    """networks.optimizers.Adam on a flattened decoder: with the fused kernel and with DPF_FUSED_ADAM=0 (the op sequence) the
    weights, moments and step counters after 5 steps are bit-identical."""
    nets = _gpu()
    import copy
    from dpf_nets_amd.networks import optimizers as O
    torch.manual_seed(3)
    base = nets.LocalCondRNVPDecoder(1, 64, 32).cuda().train()
    res = []
    for fused in (True, False):
        monkeypatch.setattr(O, "FUSED_ADAM", fused)
        dec = copy.deepcopy(base)
        store = dec.flatten_parameters()
        opt = nets.Adam(list(dec.parameters()), lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-3, amsgrad=True)
        gen = torch.Generator(device="cuda").manual_seed(9)
        for _ in range(5):
            store.flat_g.copy_(torch.randn(store.flat_g.numel(), device="cuda", generator=gen))
            store.grad_written = True
            opt.step()
        w = dec.flows[0].nvp2.T_mu_0[3].weight
        res.append((store.flat_p.clone(), opt.state[w]["exp_avg"].clone(), opt.state[w]["max_exp_avg_sq"].clone(), opt.state[w]["step"]))
    for a, b in zip(*res):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
