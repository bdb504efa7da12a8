"""Host driver of the training-mode HIP flow path (csrc/flow_train.hip via include/dpf_hip.h).

model.train() puts BatchNorm batch statistics and autograd on the decoder path
(lib/networks/flows.py:95-117, lib/networks/training.py:55).  Here the per-point work of every
coupling layer -- forward with batch statistics AND the whole backward pass -- runs in the HIP
kernels; torch is the plumbing around them:

  * `stack_parameters` gathers the layers' parameters into the differentiable (L, 8968) block
    the C ABI takes, so the kernels' parameter gradients flow back to the nn.Parameters through
    one torch.cat;
  * `film_vectors` evaluates the 4L per-cloud FiLM conditioner nets (B x 64 tensors, batch-stat
    BatchNorm over the B clouds) batched as two bmm's on PyTorch-ROCm -- they are O(B), not
    O(B*N), and their autograd graph is kept;
  * `_FlowStackTrain` is ONE autograd node for the L-layer stack: forward = per layer
    dpf_flow_train_prepare_layer + dpf_flow_forward(n_layers=1); backward = per layer
    dpf_flow_train_backward_layer in reverse order.
"""
import os

import torch

from .._lib import lib, check, current_stream, PREC, MODE
from .engine import layer_meta, _pad_cols, _pad_rows

F = 64
# precision of the forward contraction and of its recomputation in the backward passes (which fixes
# every ReLU mask): bf16x6 is fp32-class; bf16x3 (~1e-5) is ~25 % faster and flips ~1e-5 of the ReLUs
TRAIN_PRECISION = os.environ.get("DPF_TRAIN_PRECISION", "bf16x6")
BRANCHES = ("logvar", "mu")
SUBS = ("w", "b")


def stack_parameters(layers):
    """(L, dpf_flow_train_canon_floats) differentiable parameter block, layout of dpf_hip.h."""
    rows = []
    for lyr in layers:
        pieces = []
        for br in BRANCHES:
            t0 = getattr(lyr, "T_%s_0" % br)
            sd0, bn0, sd1 = t0[0], t0[1], t0[3]
            sd2 = getattr(lyr, "T_%s_1" % br)[1]
            b2 = sd2.bias[0]
            pieces += [_pad_cols(sd0.weight[0], 2).reshape(-1), bn0.weight, bn0.bias, sd1.weight[0].reshape(-1),
                       _pad_rows(sd2.weight[0], 2).reshape(-1), torch.cat([b2, b2.new_zeros(4 - b2.shape[0])])]
        rows.append(torch.cat(pieces))
    return torch.stack(rows)


def _film_modules(layers):
    return [getattr(lyr, "T_%s_0_cond_%s" % (br, s)) for lyr in layers for br in BRANCHES for s in SUBS]


def film_vectors(layers, g, update_stats=True):
    """FiLM vectors of all layers: (L, 2 branches, 2 (w|b), B, 64), differentiable w.r.t. g and the
    conditioner parameters.  Linear -> BatchNorm1d (batch statistics over the B clouds) -> Swish ->
    Linear (flows.py:33-45, 68-80)."""
    mods = _film_modules(layers)
    B = g.shape[0]
    W0 = torch.stack([m[0].weight for m in mods])                       # (K, 64, G)
    gam = torch.stack([m[1].weight for m in mods]).unsqueeze(1)         # (K, 1, 64)
    bet = torch.stack([m[1].bias for m in mods]).unsqueeze(1)
    W1 = torch.stack([m[3].weight for m in mods])                       # (K, 64, 64)
    b1 = torch.stack([m[3].bias for m in mods]).unsqueeze(1)
    u = torch.matmul(g.unsqueeze(0), W0.transpose(1, 2))                # (K, B, 64)
    if B < 2:
        raise ValueError("Expected more than 1 value per channel when training")     # as nn.BatchNorm1d
    mean = u.mean(1, keepdim=True)
    var = u.var(1, unbiased=False, keepdim=True)
    bn_eps = mods[0][1].eps
    y = (u - mean) * torch.rsqrt(var + bn_eps) * gam + bet
    y = y * torch.sigmoid(y)
    out = torch.matmul(y, W1.transpose(1, 2)) + b1
    if update_stats:
        with torch.no_grad():
            bns = [m[1] for m in mods]
            _update_running(bns, list(mean.detach().squeeze(1).unbind(0)),
                            list((var.detach().squeeze(1) * (B / (B - 1.0))).unbind(0)))
    return out.view(len(layers), 2, 2, B, F)


def _update_running(bns, means, uvars):
    """running = (1 - momentum) * running + momentum * batch  (nn.BatchNorm1d, momentum 0.1)"""
    m = bns[0].momentum
    rms, rvs, nbt = [b.running_mean for b in bns], [b.running_var for b in bns], [b.num_batches_tracked for b in bns]
    torch._foreach_mul_(rms, 1.0 - m)
    torch._foreach_add_(rms, means, alpha=m)
    torch._foreach_mul_(rvs, 1.0 - m)
    torch._foreach_add_(rvs, uvars, alpha=m)
    torch._foreach_add_(nbt, 1)


class _FlowStackTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, tcanon, fm, metas, mode, eps, prec):
        L = tcanon.shape[0]
        B, _, N = p.shape
        dev = p.device
        L_ = lib()
        stream = current_stream()
        p = p.contiguous()
        tcanon = tcanon.contiguous()
        fm = fm.contiguous()
        packed = torch.empty(L_.dpf_flow_train_packed_bytes(L, prec), dtype=torch.uint8, device=dev)
        check(L_.dpf_flow_train_pack(L, prec, tcanon.data_ptr(), packed.data_ptr(), stream), "flow_train_pack")
        pbytes = L_.dpf_flow_train_packed_bytes(1, prec)
        film = torch.empty((L, L_.dpf_flow_train_film_floats(B)), dtype=torch.float32, device=dev)
        stats = torch.empty((L, L_.dpf_flow_train_stats_floats()), dtype=torch.float32, device=dev)
        ws = torch.empty(L_.dpf_flow_train_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
        meta_dev = torch.tensor(metas, dtype=torch.int32, device=dev)
        ps, mus, lvs = (torch.empty((L, B, 3, N), dtype=torch.float32, device=dev) for _ in range(3))
        order = list(range(L)) if mode == "direct" else list(range(L - 1, -1, -1))
        cur = p
        for l in order:
            ka, kb, wa, wb = metas[l]
            pk = packed.data_ptr() + l * pbytes
            check(L_.dpf_flow_train_prepare_layer(B, N, prec, ka, kb, tcanon[l].data_ptr(), pk, fm[l].data_ptr(),
                                                  cur.data_ptr(), stats[l].data_ptr(), film[l].data_ptr(), eps,
                                                  ws.data_ptr(), stream), "flow_train_prepare_layer")
            check(L_.dpf_flow_forward(1, B, N, MODE[mode], prec, pk, meta_dev[l].data_ptr(),
                                      film[l].data_ptr(), cur.data_ptr(), ps[l].data_ptr(), None, None,
                                      ps[l].data_ptr(), mus[l].data_ptr(), lvs[l].data_ptr(), eps, stream),
                  "flow_forward")
            cur = ps[l]
        ctx.save_for_backward(p, tcanon, packed, film, stats, ps)
        ctx.metas, ctx.mode, ctx.eps, ctx.order, ctx.prec = metas, mode, eps, order, prec
        ctx.mark_non_differentiable(stats)
        return ps, mus, lvs, stats

    @staticmethod
    def backward(ctx, g_ps, g_mus, g_lvs, _g_stats):
        p, tcanon, packed, film, stats, ps = ctx.saved_tensors
        L = tcanon.shape[0]
        B, _, N = p.shape
        dev = p.device
        L_ = lib()
        stream = current_stream()
        pbytes = L_.dpf_flow_train_packed_bytes(1, ctx.prec)
        ws = torch.empty(L_.dpf_flow_train_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
        dcanon = torch.empty_like(tcanon)
        dfm = torch.empty((L, 2, 2, B, F), dtype=torch.float32, device=dev)
        g_ps, g_mus, g_lvs = g_ps.contiguous(), g_mus.contiguous(), g_lvs.contiguous()
        dp = [torch.empty_like(p), torch.empty_like(p)]
        chain = None
        order = ctx.order
        for step in range(L - 1, -1, -1):
            l = order[step]
            p_in = p if step == 0 else ps[order[step - 1]]
            ka, kb, wa, wb = ctx.metas[l]
            out = dp[step & 1]
            check(L_.dpf_flow_train_backward_layer(
                B, N, MODE[ctx.mode], ctx.prec, ka, kb, wa, wb, tcanon[l].data_ptr(), packed.data_ptr() + l * pbytes,
                film[l].data_ptr(), stats[l].data_ptr(), p_in.data_ptr(), g_ps[l].data_ptr(),
                chain.data_ptr() if chain is not None else None, g_mus[l].data_ptr(), g_lvs[l].data_ptr(),
                out.data_ptr(), dcanon[l].data_ptr(), dfm[l].data_ptr(), ctx.eps, ws.data_ptr(), stream),
                "flow_train_backward_layer")
            chain = out
        return chain, dcanon, dfm, None, None, None, None


def run_training_stack(layers, p, g, mode, precision=None):
    """Training-mode forward of `layers` (DIRECT order) on the HIP path.  Returns (ps, mus, lvs):
    three (L,B,3,N) tensors in DIRECT order, attached to autograd; updates the BatchNorm running
    statistics as nn.BatchNorm1d would."""
    if not p.is_cuda or not g.is_cuda:
        raise RuntimeError("the HIP training path runs on MI355X only (p and g must be CUDA tensors)")
    if p.dtype != torch.float32 or g.dtype != torch.float32:
        raise RuntimeError("p and g must be float32")
    if p.dim() != 3 or p.shape[1] != 3 or g.dim() != 2 or g.shape[0] != p.shape[0]:
        raise RuntimeError("expected p (B,3,N) and g (B,G)")
    if layers[0].f_n_features != F:
        raise RuntimeError("dpf_hip flow kernels are built for f_n_features == 64")
    if p.shape[0] * p.shape[2] < 2:
        raise ValueError("Expected more than 1 value per channel when training")
    precision = precision or TRAIN_PRECISION
    if precision not in ("bf16x3", "bf16x6"):
        raise ValueError("training precision must be bf16x3 or bf16x6")
    with torch.cuda.device(p.device):
        tcanon = stack_parameters(layers)
        fm = film_vectors(layers, g)
        metas = tuple(tuple(layer_meta(l)) for l in layers)
        ps, mus, lvs, stats = _FlowStackTrain.apply(p, tcanon, fm, metas, mode, float(layers[0].eps_value),
                                                      PREC[precision])
        with torch.no_grad():
            st = stats[:, :2 * 6 * F].view(len(layers), 2, 6, F)
            bns, means, uvars = [], [], []
            for li, lyr in enumerate(layers):
                for bi, br in enumerate(BRANCHES):
                    t0 = getattr(lyr, "T_%s_0" % br)
                    bns += [t0[1], t0[4]]
                    means += [st[li, bi, 0], st[li, bi, 2]]
                    uvars += [st[li, bi, 4], st[li, bi, 5]]
            _update_running(bns, means, uvars)
    return ps, mus, lvs
