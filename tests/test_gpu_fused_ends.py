"""the steps either side of the point flow, fused.
  * `reparameterize` (lib/networks/models.py:76-79, :212) in the direct stack's prologue: LocalCondRNVPDecoder.sample_and_decode
  * PointFlowNLL (losses.py:11-15) of a training step as ONE autograd node over HIP kernels (forward pass + backward launch)."""
import math

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


@pytest.mark.parametrize("base", ["fixed", "per_cloud"])
def test_sample_and_decode_equals_reparameterize_then_forward(base):
    nets = _gpu()
    B, N, G, nf = 5, 700, 128, 2
    dec = nets.LocalCondRNVPDecoder(nf, 64, G)
    dec.load_state_dict(FO.to_torch(FO.make_decoder_state(21, nf, 64, G)), strict=True)
    dec = dec.cuda().eval()
    _, _, g = FO.synthetic_inputs(21, B, N, G)
    tg = torch.from_numpy(g).cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    if base == "fixed":                                     # models.py:203-209: buffers (1,3,1) expanded over batch and points
        mu0 = torch.zeros(1, 3, 1, device="cuda").expand(B, 3, N)
        lv0 = (-3.596 * torch.ones(1, 3, 1, device="cuda")).expand(B, 3, N)
    else:                                                   # models.py:188-201: per-cloud (B,3) vectors expanded over the points
        mu0 = (0.1 * torch.randn(B, 3, device="cuda", generator=gen)).unsqueeze(2).expand(B, 3, N)
        lv0 = (-3.0 + 0.3 * torch.randn(B, 3, device="cuda", generator=gen)).unsqueeze(2).expand(B, 3, N)
    noise = torch.randn(B, 3, N, device="cuda", generator=gen)
    with torch.no_grad():
        z_ref = noise.mul(torch.exp(0.5 * lv0)).add_(mu0)                   # models.py:77-79
        ps_r, mus_r, lvs_r = dec(z_ref, tg, mode="direct")
        z, ps, mus, lvs = dec.sample_and_decode(mu0, lv0, tg, noise=noise)
    assert torch.allclose(z, z_ref, rtol=1e-6, atol=1e-8)
    assert len(ps) == len(ps_r) == 3 * nf
    for k in (0, 3, 3 * nf - 1):
        assert torch.allclose(ps[k], ps_r[k], rtol=1e-5, atol=1e-6) and torch.allclose(lvs[k], lvs_r[k], rtol=1e-5, atol=1e-7)
    assert torch.allclose(lvs.total(), lvs_r.total(), rtol=1e-5, atol=1e-6)
    # the caller's list handling (models.py:211-216) and the evaluation NLL over the fused result
    samples = [z]; samples += ps
    m = [mu0]; m += mus
    lv = [lv0]; lv += lvs
    nll = nets.PointFlowNLL()(samples, m, lv)
    nll_r = nets.PointFlowNLL()([z_ref] + list(ps_r), [mu0] + list(mus_r), [lv0] + list(lvs_r))
    assert abs(float(nll) - float(nll_r)) <= 2e-6 * abs(float(nll_r))
    # default noise: drawn by torch.randn_like -> the caller's generator stream (same seed, same draw)
    torch.manual_seed(11)
    with torch.no_grad():
        z1 = dec.sample_and_decode(mu0, lv0, tg)[0]
    torch.manual_seed(11)
    z2 = torch.randn_like(lv0).mul(torch.exp(0.5 * lv0)).add_(mu0)
    assert torch.allclose(z1, z2, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("base", ["fixed", "learned"])
def test_training_nll_is_one_hip_node_with_the_tensor_op_gradients(base):
    nets = _gpu()
    from dpf_nets_amd.networks import losses
    B, N, G, nf = 6, 300, 128, 1
    sd = FO.to_torch(FO.make_decoder_state(5, nf, 64, G))
    tgt, _, g = FO.synthetic_inputs(5, B, N, G)
    res = {}
    for impl in ("fused", "tensor"):
        dec = nets.LocalCondRNVPDecoder(nf, 64, G)
        dec.load_state_dict(sd, strict=True)
        dec = dec.cuda().train()
        tp = torch.from_numpy(tgt.copy()).cuda().requires_grad_(True)
        tg = torch.from_numpy(g.copy()).cuda().requires_grad_(True)
        lvp = torch.full((B, 3), -3.6, device="cuda").add_(0.1 * torch.arange(B * 3, device="cuda").reshape(B, 3) / (B * 3))
        lvp.requires_grad_(base == "learned")
        mu0 = torch.zeros(1, 3, 1, device="cuda").expand(B, 3, N)
        lv0 = lvp.unsqueeze(2).expand(B, 3, N)                               # models.py:156-158 ('freevar')
        ps, mus, lvs = dec(tp, tg, mode="inverse")
        samples, m, lv = ps + [tp], [mu0] + mus, [lv0] + lvs                  # models.py:169-171
        if impl == "fused":
            loss = nets.PointFlowNLL()(samples, m, lv)
            assert type(loss.grad_fn).__name__ == "_PointFlowNLLNodeBackward"    # ONE node, no elementwise graph
        else:
            tot = sum(lv) + (samples[0] - m[0]) ** 2 / torch.exp(lv[0])      # losses.py:13, as tensor ops
            loss = 0.5 * (tot.sum() / B + math.log(2.0 * math.pi) * 3 * N)
        loss.backward()
        res[impl] = dict(loss=float(loss), gp=tp.grad.clone(), gg=tg.grad.clone(), glv=None if lvp.grad is None else lvp.grad.clone(),
                         gw={k: v.grad.clone() for k, v in dec.named_parameters()})
    a, b = res["fused"], res["tensor"]
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(b["loss"])
    for k in ("gp", "gg"):
        assert torch.allclose(a[k], b[k], rtol=2e-4, atol=1e-6 * float(b[k].abs().max())), k
    if base == "learned":
        assert torch.allclose(a["glv"], b["glv"], rtol=1e-4, atol=1e-6 * float(b["glv"].abs().max()))
    for k in b["gw"]:
        assert torch.allclose(a["gw"][k], b["gw"][k], rtol=2e-3, atol=2e-5 * float(b["gw"][k].abs().max()) + 1e-9), k
    assert losses._PointFlowNLLNode is not None
