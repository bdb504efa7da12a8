"""Parity of the TRAINING-mode PointNet encoder on csrc/encoder_train.hip (batch-statistics BatchNorm, max over the
points, backward to the twelve parameter gradients; through the C ABI via PointNetCloudEncoder.train()) against

  * the golden vectors captured from the reference's PointNetCloudEncoder in train() mode + torch.max + backward
    (encoders.py:9-28, models.py:131, training.py:55) by oracle/gen_golden.py: pooled features, projections of every
    parameter gradient, BatchNorm running statistics;
  * the same module evaluated with float64 tensor ops on the same seeded inputs, at shapes with ragged tiles, partial
    and many workgroups.

Tolerances (north star: <= 1e-4 relative to the tensor's scale for bf16x3): written out below."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from oracle import detrng
from oracle import encoder_oracle as EO
from oracle import flow_oracle as FO
from oracle.gen_golden import _grad_projection

pytestmark = pytest.mark.gpu

TOL_OUT = 1e-4          # pooled features, relative to the tensor's largest magnitude
TOL_GRAD = 5e-4         # parameter gradients, relative to the gradient tensor's largest magnitude
TOL_STAT = 2e-5         # running statistics


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd import networks
    return networks


def rel(got, ref):
    got = got.detach().double().cpu().numpy()
    ref = ref.detach().double().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def _encoder(nets, seed):
    enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512])
    enc.load_state_dict(FO.to_torch(EO.make_encoder_state(seed)), strict=True)
    return enc.cuda().train()


def _step(enc, x, r):
    for p in enc.parameters():
        p.grad = None
    x.grad = None
    feats = enc(x)
    pooled = torch.max(feats, dim=2)[0]
    (pooled * r).sum().backward()
    return feats, pooled


def test_encoder_train_vs_reference_golden(golden_dir):
    nets = _gpu()
    gold = np.load(os.path.join(golden_dir, "encoder.npz"))
    meta = json.load(open(os.path.join(golden_dir, "encoder.json")))
    for case, (seed, B, N) in meta["cases"].items():
        enc = _encoder(nets, seed)
        x = torch.from_numpy(EO.encoder_inputs(seed, B, N)).cuda().requires_grad_(True)     # the golden run took d/dx too
        r = torch.from_numpy(detrng.normal_f32(detrng.key(seed, "enc_r"), (B, 512))).cuda()
        feats, pooled = _step(enc, x, r)
        assert isinstance(feats, nets.TrainPointFeatures) and feats._full is None      # no (B,512,N) tensor was formed
        tag = case + "_train"
        assert rel(pooled, gold[tag + "_max"]) <= TOL_OUT, (case, rel(pooled, gold[tag + "_max"]))
        assert rel(x.grad, gold[tag + "_dx"]) <= TOL_GRAD, (case, rel(x.grad, gold[tag + "_dx"]))
        for k, v in _grad_projection([(k, p.grad.cpu()) for k, p in enc.named_parameters()], seed).items():
            ref = gold[tag + "_gproj_" + k]
            for i in range(3):
                assert abs(v[i] - ref[i]) <= 1e-3 * (ref[2] + 1e-6) + 1e-5, (case, k, v, ref)
        for k, v in enc.state_dict().items():
            if "running" in k:
                assert rel(v, gold[tag + "_stat_" + k]) <= TOL_STAT, (case, k, rel(v, gold[tag + "_stat_" + k]))
            if "num_batches" in k:
                assert int(v) == 1


@pytest.mark.parametrize("B,N", [(2, 5), (2, 31), (3, 33), (2, 255), (1, 257), (5, 700), (4, 2048), (8, 20011)])
def test_encoder_train_vs_float64_tensor_ops(B, N):
    nets = _gpu()
    enc = _encoder(nets, 300 + N)
    ref = copy.deepcopy(enc).double()
    ref.hip_training = False
    x = torch.from_numpy(EO.encoder_inputs(400 + N, B, N)).cuda().requires_grad_(True)
    r = torch.from_numpy(detrng.normal_f32(detrng.key(N, "r"), (B, 512))).cuda()
    feats, pooled = _step(enc, x, r)
    assert isinstance(feats, nets.TrainPointFeatures)
    x64 = x.detach().double().requires_grad_(True)
    rfeats, rpooled = _step(ref, x64, r.double())
    assert torch.is_tensor(rfeats)
    assert rel(pooled, rpooled) <= TOL_OUT, rel(pooled, rpooled)
    assert rel(x.grad, x64.grad) <= TOL_GRAD, rel(x.grad, x64.grad)
    for (k, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and p.grad.shape == p.shape
        assert rel(p.grad, q.grad) <= TOL_GRAD, (k, rel(p.grad, q.grad))
    for (k, v), (_, u) in zip(enc.state_dict().items(), ref.state_dict().items()):
        if "running" in k:
            assert rel(v, u) <= TOL_STAT, (k, rel(v, u))
        if "num_batches" in k:
            assert int(v) == int(u) == 1


def test_encoder_train_deterministic_and_semantics():
    """Two runs give bitwise-identical outputs and gradients; statistics are taken once per call whichever of the
    features' uses comes first; a differentiable input, hip_training = False and no_grad take their documented paths."""
    nets = _gpu()
    B, N = 4, 1000
    x = torch.from_numpy(EO.encoder_inputs(7, B, N)).cuda()
    r = torch.from_numpy(detrng.normal_f32(detrng.key(7, "r"), (B, 512))).cuda()
    runs = []
    for _ in range(2):
        enc = _encoder(nets, 8)
        _, pooled = _step(enc, x, r)
        runs.append([pooled.detach().clone()] + [p.grad.clone() for p in enc.parameters()] +
                    [v.clone() for k, v in enc.state_dict().items() if "running" in k])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # max first, then the features: running statistics move once, and both agree
    enc = _encoder(nets, 8)
    feats = enc(x)
    pooled = torch.max(feats, dim=2)[0]
    rm = enc.features.sd2_bn.running_mean.clone()
    full = feats.tensor()
    assert torch.equal(enc.features.sd2_bn.running_mean, rm) and int(enc.features.sd2_bn.num_batches_tracked) == 1
    assert rel(pooled, full.max(dim=2)[0]) <= TOL_OUT
    # features first: tensor ops throughout
    enc = _encoder(nets, 8)
    feats = enc(x)
    full = feats * 1.0
    assert torch.is_tensor(full) and full.requires_grad and int(enc.features.sd2_bn.num_batches_tracked) == 1
    assert rel(torch.max(feats, dim=2)[0], pooled) <= TOL_OUT and int(enc.features.sd2_bn.num_batches_tracked) == 1
    # no_grad in training mode: forward only, statistics still move
    enc = _encoder(nets, 8)
    with torch.no_grad():
        p2 = torch.max(enc(x), dim=2)[0]
    assert torch.equal(p2, pooled) and not p2.requires_grad and int(enc.features.init_sd_bn.num_batches_tracked) == 1
    # the eval-mode kernel sees the running statistics the training pass wrote (version bump)
    enc.eval()
    with torch.no_grad():
        e1 = torch.max(enc(x), dim=2)[0]
        e2 = torch.max(enc.forward_torch(x), dim=2)[0]
    assert rel(e1, e2) <= TOL_OUT
    assert torch.is_tensor(enc(x.clone().requires_grad_(True)))      # eval mode, differentiable input: tensor ops
    enc.train()
    enc.hip_training = False
    assert torch.is_tensor(enc(x))


def test_encoder_train_full_size_error_vs_float64():
    """cfg-2 size (B=32, N=2048): error of the HIP gradients against float64, next to the error of the fp32 tensor-op
    path against the same float64 reference."""
    nets = _gpu()
    B, N = 32, 2048
    enc = _encoder(nets, 21)
    ref = copy.deepcopy(enc).double()
    ref.hip_training = False
    t32 = copy.deepcopy(enc)
    t32.hip_training = False
    x = torch.from_numpy(EO.encoder_inputs(22, B, N)).cuda()
    r = torch.from_numpy(detrng.normal_f32(detrng.key(22, "r"), (B, 512))).cuda()
    _, pooled = _step(enc, x, r)
    _, rpooled = _step(ref, x.double(), r.double())
    _, tpooled = _step(t32, x, r)
    assert rel(pooled, rpooled) <= TOL_OUT
    worst = 0.0
    for (k, p), (_, q), (_, t) in zip(enc.named_parameters(), ref.named_parameters(), t32.named_parameters()):
        e_hip, e_t32 = rel(p.grad, q.grad), rel(t.grad, q.grad)
        worst = max(worst, e_hip)
        assert e_hip <= TOL_GRAD, (k, e_hip, e_t32)
    print("worst relative gradient error vs float64: %.3g" % worst)
