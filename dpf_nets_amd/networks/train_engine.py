"""Host driver of the training-mode HIP flow path (csrc/flow_train.hip via include/dpf_hip.h).

model.train() puts BatchNorm batch statistics and autograd on the decoder path
(lib/networks/flows.py:95-117, lib/networks/training.py:55).  Here the per-point work of every
coupling layer -- forward with batch statistics AND the whole backward pass -- runs in the HIP
kernels, and the whole L-layer stack is ONE autograd node (`_FlowStackTrain`) whose inputs are
p, g and every parameter of the layers:

  forward   gather the conditioner parameters into the (L, 8968) block of the C ABI with one
            torch.cat (each parameter appears flattened exactly as stored, see dpf_hip.h);
            evaluate the 4L per-cloud FiLM nets (B x 64 tensors; batch-stat BatchNorm over the B
            clouds): one launch of dpf_film_train_forward (r03; B > 64: two bmm's and a dozen tensor ops);
            per layer dpf_flow_train_prepare_layer + dpf_flow_forward(n_layers=1);
  backward  per layer dpf_flow_train_backward_layer in reverse order; the FiLM nets' backward as one launch of
            dpf_film_train_backward (adding into the flat store) or batched tensor ops; one multi-tensor copy hands
            every parameter its gradient (per-parameter path).

A model of n_flows=21 has 2016 parameter tensors on this path; letting autograd slice, stack and
accumulate them one by one costs more host time than all the kernels together, which is why the
node owns them all.
"""
import ctypes
import os

import torch

from .._lib import lib, check, current_stream, PREC, MODE
from .engine import layer_meta

F = 64
# precision of the forward contraction and of its recomputation in the backward passes (which fixes
# every ReLU mask).  f16x3 (default since r03): fp16 hi + fp16 lo operands, three products -- fp32-class (22 significant
# bits) at half the matrix work of bf16x6, exact while the post-BN0 activations stay below 2048 (the lo part of the split is
# clamped to [0, 1], csrc/flow_common.h split_relu_f16); a per-stack monitor of max|gamma0| * sqrt(B*N) + max|beta0| (the
# bound of a batch-normalised activation) falls back to bf16x6 for good when that bound is reached (F16_LIMIT).
# bf16x6 is fp32-class without a range limit; bf16x3 (~1e-5) flips ~1e-5 of the ReLUs.
TRAIN_PRECISION = os.environ.get("DPF_TRAIN_PRECISION", "f16x3")
F16_LIMIT = 2048.0
F16_CHECK_EVERY = 16
BRANCHES = ("logvar", "mu")
SUBS = ("w", "b")

# canonical block offsets (floats) per branch, dpf_hip.h
_T_W0, _T_G0, _T_B0, _T_W1, _T_W2, _T_B2, _T_BR = 0, 128, 192, 256, 4352, 4480, 4484


class StackSpec:
    """Static description of a list of coupling layers: which parameter goes where."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.L = len(self.layers)
        self.metas = tuple(tuple(layer_meta(l)) for l in self.layers)
        self.G = self.layers[0].g_n_features
        self.eps = float(self.layers[0].eps_value)
        self._dev_cache = {}
        self._params = None
        self.flat = None               # FlatStore once flatten() was called
        self.f16_ok = True             # f16x3 range monitor (below): False once the activation bound reached F16_LIMIT
        self._f16_calls = 0
        self._f16_pending = None       # (pinned host tensor, event) of a bound computed some calls ago
        self.meta_host = (ctypes.c_int * (4 * self.L))(*[v for m in self.metas for v in m])
        self.canon_slots = []          # (offset, numel) per parameter, in canon_params() order
        pads = []                      # (offset, numel) of the zero padding between them
        for li, lyr in enumerate(self.layers):
            nk, nw = len(lyr.keep_inds), len(lyr.warp_inds)
            for bi in range(2):
                base = (li * 2 + bi) * _T_BR
                self.canon_slots += [(base + _T_W0, 64 * nk), (base + _T_G0, 64), (base + _T_B0, 64),
                                     (base + _T_W1, 4096), (base + _T_W2, 64 * nw), (base + _T_B2, nw)]
                if nk < 2:
                    pads.append((base + _T_W0 + 64, 64))
                if nw < 2:
                    pads.append((base + _T_W2 + 64, 64))
                pads.append((base + _T_B2 + nw, 4 - nw))
        # the gather is a single cat over parameters and zero pads in offset order
        order = sorted([(o, n, ("p", i)) for i, (o, n) in enumerate(self.canon_slots)] + [(o, n, ("z", n)) for (o, n) in pads])
        pos = 0
        for o, n, _ in order:
            assert o == pos, (o, pos)
            pos += n
        assert pos == self.L * 2 * _T_BR
        self.cat_plan = [what for _, _, what in order]

    def flatten(self, dev):
        """Move the parameters and BatchNorm buffers of the layers into one FlatStore (idempotent)."""
        if self.flat is None or not self.flat.attached():
            self.flat = FlatStore(self, dev)
        return self.flat

    def f16_range_monitor(self, tcanon, count):
        """(see _f16_monitor) tcanon: the (L, 2*T_BR) conditioner block, or None = take gamma0 / beta0 from the modules."""
        def bound():
            root = float(count) ** 0.5
            if tcanon is not None:
                blk = tcanon.detach().view(self.L, 2, _T_BR)
                return blk[:, :, _T_G0:_T_G0 + 64].abs().amax() * root + blk[:, :, _T_B0:_T_B0 + 64].abs().amax()
            bns = [getattr(lyr, "T_%s_0" % br)[1] for lyr in self.layers for br in BRANCHES]
            g = torch.stack([b.weight.detach().abs().amax() for b in bns]).amax()
            return g * root + torch.stack([b.bias.detach().abs().amax() for b in bns]).amax()
        return self._f16_monitor(bound)

    def _f16_monitor(self, bound_fn):
        """Non-blocking guard of the f16x3 split's exact range.  Every F16_CHECK_EVERY-th call the bound
        max|gamma0| * sqrt(count) + max|beta0| over all layers (|BN0(h0)| cannot exceed it: a standardised value of `count`
        samples is at most sqrt(count - 1)) is computed on the device and copied to pinned memory; a later call reads it
        once the copy's event has completed -- no host synchronisation on the training path.  Parameters move by ~lr per
        step, so a lag of a few steps is immaterial next to a limit of 2048.  Returns the precision to use NOW."""
        if not self.f16_ok:
            return "bf16x6"
        if torch.cuda.is_current_stream_capturing():
            # the caller is capturing this step into a torch.cuda.graph: an event query is illegal during capture, an event
            # recorded inside it cannot be queried by a later eager call, and a replay re-runs none of this host code -- the
            # precision chosen now is frozen into the graph.  Neither record nor query; a bound that was requested before
            # the capture began stays pending for the next eager call.  (Under whole-step capture run one eager step every
            # few hundred replays so that the range guard still sees the weights: INTEGRATION.md.)
            return "f16x3"
        pend = self._f16_pending
        if pend is not None and pend[1].query():
            bound = float(pend[0][0])
            self._f16_pending = None
            if not (bound < F16_LIMIT):            # also catches NaN
                import warnings
                self.f16_ok = False
                warnings.warn("dpf_nets_amd: post-BatchNorm activation bound %.3g reached the f16x3 split's exact range (%g); "
                              "this stack trains at bf16x6 from now on" % (bound, F16_LIMIT))
                return "bf16x6"
        if self._f16_pending is None and self._f16_calls % F16_CHECK_EVERY == 0:
            with torch.no_grad():
                bound = bound_fn()
                host = self._dev_cache.get("f16_pinned")
                if host is None:
                    host = self._dev_cache["f16_pinned"] = torch.empty(1, dtype=torch.float32, pin_memory=True)
                host.copy_(bound.reshape(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._f16_pending = (host, ev)
        self._f16_calls += 1
        return "f16x3"

    def meta_on(self, dev):
        """int32 (L,4) keep/warp table on the device (cached: a host->device copy synchronises)."""
        t = self._dev_cache.get(("meta", dev))
        if t is None:
            t = self._dev_cache[("meta", dev)] = torch.tensor(self.metas, dtype=torch.int32, device=dev)
        return t

    def zeros_on(self, dev):
        t = self._dev_cache.get(("zeros", dev))
        if t is None:
            t = self._dev_cache[("zeros", dev)] = torch.zeros(64, dtype=torch.float32, device=dev)
        return t

    def canon_params(self):
        out = []
        for lyr in self.layers:
            for br in BRANCHES:
                t0 = getattr(lyr, "T_%s_0" % br)
                sd2 = getattr(lyr, "T_%s_1" % br)[1]
                out += [t0[0].weight, t0[1].weight, t0[1].bias, t0[3].weight, sd2.weight, sd2.bias]
        return out

    def film_modules(self):
        return [getattr(lyr, "T_%s_0_cond_%s" % (br, s)) for lyr in self.layers for br in BRANCHES for s in SUBS]

    def film_params(self):
        out = []
        for m in self.film_modules():
            out += [m[0].weight, m[1].weight, m[1].bias, m[3].weight, m[3].bias]
        return out

    def all_params(self):
        """canon_params() + film_params(), cached: walking 2016 nn.Sequential items costs ~4 ms per step.  The
        Parameter OBJECTS survive .to()/.cuda(), load_state_dict and optimizer steps (all in place); the two
        sentinels catch a parameter that was re-assigned."""
        c = self._params
        lyr0, lyrN = self.layers[0], self.layers[-1]
        if c is None or c[0] is not lyr0.T_logvar_0[0].weight or c[-1] is not lyrN.T_mu_0_cond_b[3].bias:
            c = self._params = self.canon_params() + self.film_params()
        return c

    def flow_bns(self):
        out = []
        for lyr in self.layers:
            for br in BRANCHES:
                t0 = getattr(lyr, "T_%s_0" % br)
                out += [t0[1], t0[4]]
        return out


def _update_running(bns, means, uvars):
    """running = (1 - momentum) * running + momentum * batch  (nn.BatchNorm1d, momentum 0.1)"""
    m = bns[0].momentum
    rms, rvs, nbt = [b.running_mean for b in bns], [b.running_var for b in bns], [b.num_batches_tracked for b in bns]
    torch._foreach_mul_(rms, 1.0 - m)
    torch._foreach_add_(rms, means, alpha=m)
    torch._foreach_mul_(rvs, 1.0 - m)
    torch._foreach_add_(rvs, uvars, alpha=m)
    torch._foreach_add_(nbt, 1)


def _scatter(shapes_like, flat_views):
    """Fresh gradient tensors (autograd takes them over without a copy) filled by one multi-tensor copy."""
    outs = [torch.empty_like(t) for t in shapes_like]
    torch._foreach_copy_(outs, [v.view_as(o) for v, o in zip(flat_views, outs)])
    return outs


def _forward_core(p, g, spec, mode, prec, tcanon, W0, gam, bet, W1, b1, flat=None):
    """Forward of the stack given the (L, 2*T_BR) conditioner block and the batched FiLM-net weights.
    Returns (ps, mus, lvs) and the tensors the backward needs."""
    L, G = spec.L, spec.G
    B, _, N = p.shape
    dev = p.device
    L_ = lib()
    stream = current_stream()
    K = 4 * L
    if B < 2:
        raise ValueError("Expected more than 1 value per channel when training")      # as nn.BatchNorm1d
    mods = spec.film_modules()
    # ---- FiLM conditioner nets, all K = 4L of them (flows.py:33-45, 68-80)
    if film_fused_ok(B, G, W0, gam, bet, W1, b1):
        # r03: ONE launch (csrc/film_train.hip) instead of ~12 tensor ops; the backward recomputes y / sigmoid / swish
        fm = torch.empty((L, 2, 2, B, F), dtype=torch.float32, device=dev)
        xhat = torch.empty((K, B, F), dtype=torch.float32, device=dev)
        rstd, film_mean, film_uvar = (torch.empty((K, F), dtype=torch.float32, device=dev) for _ in range(3))
        check(L_.dpf_film_train_forward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(),
                                        b1.data_ptr(), float(mods[0][1].eps), fm.data_ptr(), xhat.data_ptr(), rstd.data_ptr(),
                                        film_mean.data_ptr(), film_uvar.data_ptr(), stream), "film_train_forward")
        y = sig = sw = None
    else:                                                              # batched tensor ops (B > 64, or non-contiguous blocks)
        u = torch.matmul(g.unsqueeze(0), W0.transpose(1, 2))           # (K, B, F)
        var, mean = torch.var_mean(u, dim=1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + mods[0][1].eps)
        xhat = (u - mean) * rstd
        y = xhat * gam + bet
        sig = torch.sigmoid(y)
        sw = y * sig
        fm = torch.baddbmm(b1, sw, W1.transpose(1, 2)).view(L, 2, 2, B, F).contiguous()
        film_mean, film_uvar = mean.view(K, F), var.view(K, F) * (B / (B - 1.0))
    # ---- the layers
    packed = torch.empty(L_.dpf_flow_train_packed_bytes(L, prec), dtype=torch.uint8, device=dev)
    check(L_.dpf_flow_train_pack(L, prec, tcanon.data_ptr(), packed.data_ptr(), stream), "flow_train_pack")
    film = torch.empty((L, L_.dpf_flow_train_film_floats(B)), dtype=torch.float32, device=dev)
    stats = torch.empty((L, L_.dpf_flow_train_stats_floats()), dtype=torch.float32, device=dev)
    ws = torch.empty(L_.dpf_flow_train_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
    meta_dev = spec.meta_on(dev)
    ps, mus, lvs = (torch.empty((L, B, 3, N), dtype=torch.float32, device=dev) for _ in range(3))
    check(L_.dpf_flow_train_forward(L, B, N, MODE[mode], prec, spec.meta_host, meta_dev.data_ptr(), tcanon.data_ptr(),
                                    packed.data_ptr(), fm.data_ptr(), p.data_ptr(), ps.data_ptr(), mus.data_ptr(),
                                    lvs.data_ptr(), stats.data_ptr(), film.data_ptr(), spec.eps, ws.data_ptr(), stream),
          "flow_train_forward")
    # ---- BatchNorm running statistics: FiLM nets, then the conditioner stacks
    if flat is not None and film_mean.is_contiguous() and film_uvar.is_contiguous() and flat.rm.is_contiguous() and \
            flat.nbt.dtype == torch.int64:
        flat.update_running_fused(L, film_mean, film_uvar, stats, mods[0][1].momentum)
        sv = None
    else:
        sv = stats[:, :2 * 6 * F].view(L, 2, 6, F)
        # (plain slices: `sv[:, :, (0, 2)]` is ADVANCED indexing -- it builds an index tensor on the host and copies it to the
        # device from pageable memory, which blocks the host until the stream has drained: 0.6 ms per step, r03)
        flow_mean, flow_uvar = sv[:, :, 0:3:2].reshape(4 * L, F), sv[:, :, 4:6].reshape(4 * L, F)
    if sv is None:
        pass
    elif flat is not None:
        flat.update_running(film_mean, film_uvar, flow_mean, flow_uvar, mods[0][1].momentum)
    else:
        _update_running([m[1] for m in mods], list(film_mean.unbind(0)), list(film_uvar.unbind(0)))
        _update_running(spec.flow_bns(), list(flow_mean.unbind(0)), list(flow_uvar.unbind(0)))
    # 3L output tensors (autograd hands back one gradient each) + the layer-sum of the log-variances, which is what
    # PointFlowNLL wants of them (losses.py:13): one reduction here instead of L-1 adds and L-1 backward nodes there
    outs = ps.unbind(0) + mus.unbind(0) + lvs.unbind(0) + (lvs.sum(0),)
    return outs, (p, g, tcanon, packed, film, stats, ps, W0, gam, W1, xhat, rstd, y, sig, sw, mus, lvs, bet)


def _grad_table(grads, shape, keep):
    """Host table of device pointers for L per-layer gradients (None -> NULL).  Non-contiguous gradients (the
    stride-0 expansion of a reduced loss, shared by every logvar) are materialised once per distinct tensor."""
    if all(t is None for t in grads):
        return None
    ptrs = []
    for t in grads:
        if t is None:
            ptrs.append(None)
            continue
        if not t.is_contiguous() or t.dtype != torch.float32:
            c = keep.get(id(t))
            if c is None:
                c = keep[id(t)] = t.to(torch.float32).contiguous()
            t = c
        assert tuple(t.shape) == shape
        keep.setdefault(("ref", id(t)), t)
        ptrs.append(t.data_ptr())
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def film_fused_ok(B, G, *blocks):
    """The fused FiLM-net kernels take B <= 64 clouds, G % 4 == 0 and contiguous fp32 parameter blocks."""
    return 2 <= B <= lib().dpf_film_train_max_batch() and G % 4 == 0 and \
        all(t.is_contiguous() and t.dtype == torch.float32 for t in blocks)


def _backward_core(spec, mode, prec, saved, grads, need_dg, into_flat=False):
    """grads: the 3L gradients of (ps, mus, lvs), None where an output is unused.
    -> (dL/dp, dL/dg, d canon block, dW0, dgamma, dbeta, dW1, db1), the last five batched over the K FiLM nets."""
    p, g, tcanon, packed, film, stats, ps, W0, gam, W1, xhat, rstd, y, sig, sw, mus, lvs, bet = saved
    L, G = spec.L, spec.G
    B, _, N = p.shape
    dev = p.device
    L_ = lib()
    stream = current_stream()
    ws = torch.empty(L_.dpf_flow_train_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
    dcanon = torch.empty_like(tcanon)
    dfm = torch.empty((L, 2, 2, B, F), dtype=torch.float32, device=dev)
    keep = {}
    g_lvs = grads[2 * L:3 * L]
    if len(grads) > 3 * L and grads[3 * L] is not None:        # d/d(layer-sum) reaches every layer's log-variances
        dtot = grads[3 * L]
        g_lvs = [dtot if t is None else t + dtot for t in g_lvs]
    t_ps, t_mus = (_grad_table(grads[i * L:(i + 1) * L], tuple(p.shape), keep) for i in range(2))
    t_lvs = _grad_table(g_lvs, tuple(p.shape), keep)
    chain, dp_tmp = torch.empty_like(p), torch.empty_like(p)
    check(L_.dpf_flow_train_backward_lists(L, B, N, MODE[mode], prec, spec.meta_host, tcanon.data_ptr(), packed.data_ptr(),
                                           film.data_ptr(), stats.data_ptr(), p.data_ptr(), ps.data_ptr(), mus.data_ptr(),
                                           lvs.data_ptr(), t_ps, t_mus, t_lvs,
                                           chain.data_ptr(), dp_tmp.data_ptr(), dcanon.data_ptr(), dfm.data_ptr(), spec.eps,
                                           ws.data_ptr(), stream),
          "flow_train_backward_lists")
    del keep
    # ---- FiLM nets backward
    K = 4 * L
    if y is None:                                                      # the forward took the fused kernel: so does the backward
        into = spec.flat.film_grad_blocks() if into_flat else None
        if into is None:
            dW0, dW1 = torch.empty_like(W0), torch.empty_like(W1)
            dgam, dbet, db1 = (torch.empty((K, F), dtype=torch.float32, device=dev) for _ in range(3))
        else:
            dW0, dgam, dbet, dW1, db1 = into
        dg_part = torch.empty((K, B, G), dtype=torch.float32, device=dev) if need_dg else None
        check(L_.dpf_film_train_backward(K, B, G, g.data_ptr(), W0.data_ptr(), gam.data_ptr(), bet.data_ptr(), W1.data_ptr(),
                                         xhat.data_ptr(), rstd.data_ptr(), dfm.data_ptr(), dW0.data_ptr(), dgam.data_ptr(),
                                         dbet.data_ptr(), dW1.data_ptr(), db1.data_ptr(),
                                         dg_part.data_ptr() if need_dg else None, 1 if into is not None else 0, stream),
              "film_train_backward")
        dg = dg_part.sum(0) if need_dg else None
        if into is not None:
            return chain, dg, dcanon, None, None, None, None, None
        return chain, dg, dcanon, dW0, dgam, dbet, dW1, db1
    dout = dfm.view(K, B, F)
    db1 = dout.sum(1)
    dW1 = torch.matmul(dout.transpose(1, 2), sw)                       # (K, F, F)
    dsw = torch.matmul(dout, W1)
    dy = dsw * (sig * (1.0 + y * (1.0 - sig)))
    dgam = (dy * xhat).sum(1)
    dbet = dy.sum(1)
    dxh = dy * gam
    du = rstd * (dxh - dxh.mean(1, keepdim=True) - xhat * (dxh * xhat).mean(1, keepdim=True))
    dW0 = torch.matmul(du.transpose(1, 2), g.unsqueeze(0))             # (K, F, G)
    # d g = sum_k du[k] @ W0[k].  As ONE (B, K*F) x (K*F, G) product (K*F = 16 128) the library picks a 32x32 tile kernel with
    # four workgroups that each walk the whole K: 119 us at B = 32 (tools/experiments/ae_gemm_shapes.py); as K small products
    # and a sum over K it is ~12 us
    dg = torch.bmm(du, W0).sum(0) if need_dg else None
    return chain, dg, dcanon, dW0, dgam, dbet, dW1, db1


class _FlowStackTrain(torch.autograd.Function):
    """The stack as one node whose inputs are p, g and all 32*L parameters."""

    @staticmethod
    def forward(ctx, p, g, spec, mode, prec, *params):
        L, G = spec.L, spec.G
        dev = p.device
        p = p.contiguous()
        g = g.contiguous()
        ncanon = len(spec.canon_slots)
        cparams, fparams = params[:ncanon], params[ncanon:]
        # ---- gather the conditioner parameters (one cat; zero pads come from one shared buffer)
        zeros = spec.zeros_on(dev)
        tcanon = torch.cat([cparams[i].reshape(-1) if kind == "p" else zeros[:i] for kind, i in spec.cat_plan]).view(L, 2 * _T_BR)
        K = 4 * L
        W0 = torch.cat([t.reshape(-1) for t in fparams[0::5]]).view(K, F, G)
        gam = torch.cat(fparams[1::5]).view(K, 1, F)
        bet = torch.cat(fparams[2::5]).view(K, 1, F)
        W1 = torch.cat([t.reshape(-1) for t in fparams[3::5]]).view(K, F, F)
        b1 = torch.cat(fparams[4::5]).view(K, 1, F)
        outs, saved = _forward_core(p, g, spec, mode, prec, tcanon, W0, gam, bet, W1, b1)
        ctx.save_for_backward(*saved, *params)
        ctx.spec, ctx.prec, ctx.mode = spec, prec, mode
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        saved, params = ctx.saved_tensors[:18], ctx.saved_tensors[18:]
        spec = ctx.spec
        chain, dg, dcanon, dW0, dgam, dbet, dW1, db1 = _backward_core(spec, ctx.mode, ctx.prec, saved, grads,
                                                                      ctx.needs_input_grad[1])
        # ---- hand every parameter its gradient: slices of the blocks, one multi-tensor copy
        flat = dcanon.view(-1)
        views = [flat[o:o + n] for o, n in spec.canon_slots]
        for k in range(4 * spec.L):
            views += [dW0[k], dgam[k], dbet[k], dW1[k], db1[k]]
        grads = _scatter(params, views)
        return (chain if ctx.needs_input_grad[0] else None, dg, None, None, None, *grads)


def _mark_grad_written(p):
    st = getattr(p, "_dpf_flat", None)
    if st is not None:
        st.grad_written = True


def hook_grad_written(params):
    """`grad_written` is a HINT that the store's own writers keep up to date (accumulate, attach_grads, the exchanges);
    gradients that reach flat_g by ordinary autograd -- AccumulateGrad adding in place into the attached views: the
    tensor-op path of a flattened decoder (DPF_TRAIN_IMPL=torch, forward_torch), an eval-mode decoder under autograd,
    forward(n_layers=...), a single flow module's own forward -- set it through a post-accumulate hook on every parameter
    (registered once per Parameter object; the store is looked up at call time, so a rebuilt store is found).  The hooks
    never fire on the flat path, whose node writes flat_g itself."""
    for t in params:
        if not getattr(t, "_dpf_gw_hook", False):
            t.register_post_accumulate_grad_hook(_mark_grad_written)
            t._dpf_gw_hook = True


class FlatStore:
    """ONE fp32 buffer that owns the storage of every parameter of the stack, laid out as the kernels and the batched
    FiLM ops consume it -- [conditioner block (L, 2*T_BR) | W0 (K,F,G) | gamma (K,F) | beta (K,F) | W1 (K,F,F) | b1 (K,F)] --
    a twin buffer for the gradients, and (8L, F) blocks for the BatchNorm running statistics.  Every nn.Parameter /
    buffer of the layers keeps its identity, name and shape (state dicts and optimizers are unaffected); its `.data`
    becomes a view of the flat buffer and its `.grad` a view of the gradient buffer.

    What it buys (n_flows=21: 2016 parameters, 504 BatchNorm buffers): the forward needs no gather, the backward adds
    its result blocks into the gradient buffer with six tensor ops instead of handing 2016 tensors to 2016
    AccumulateGrad nodes, and the data-parallel gradient exchange is one all-reduce of `flat_g`.

    Gradients are written by the node itself (autograd sees only p, g and a token), so per-parameter autograd hooks
    (and with them DistributedDataParallel's reducer) do not fire for these parameters: use
    dpf_nets_amd.distributed.allreduce_flat_gradients.  `optimizer.zero_grad()` may set the grads to None; the next
    backward re-attaches the views (zeroed) -- `FlatStore.zero_grad()` avoids that per-parameter pass."""

    def __init__(self, spec, dev):
        L, G = spec.L, spec.G
        K = 4 * L
        cp, fp = spec.canon_params(), spec.film_params()
        sizes = [L * 2 * _T_BR, K * F * G, K * F, K * F, K * F * F, K * F]
        shapes = [(L, 2 * _T_BR), (K, F, G), (K, 1, F), (K, 1, F), (K, F, F), (K, 1, F)]
        self.flat_p = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        self._offs, self._shapes = offs, shapes
        self.blocks = [self.flat_p[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(6)]
        self.gblocks = [self.flat_g[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(6)]
        # element range of every parameter inside the flat buffers, in spec.all_params() order
        ranges = [(o, n) for o, n in spec.canon_slots]
        for k in range(K):
            ranges += [(offs[1] + k * F * G, F * G), (offs[2] + k * F, F), (offs[3] + k * F, F),
                       (offs[4] + k * F * F, F * F), (offs[5] + k * F, F)]
        self._ranges = ranges
        self.params = cp + fp
        had_grad = any(t.grad is not None for t in self.params)
        with torch.no_grad():
            torch._foreach_copy_([self.flat_p[o:o + n] for o, n in ranges], [t.detach().reshape(-1).to(dev) for t in self.params])
        self.pviews = [self.flat_p[o:o + n].view(t.shape) for (o, n), t in zip(ranges, self.params)]
        self.gviews = [self.flat_g[o:o + n].view(t.shape) for (o, n), t in zip(ranges, self.params)]
        for t, pv, gv in zip(self.params, self.pviews, self.gviews):
            if t.grad is not None:
                gv.copy_(t.grad)
            t.data = pv
            t.grad = gv
            t._dpf_flat = self              # networks.optimizers.Adam updates a whole store at once
        hook_grad_written(self.params)
        # BatchNorm running statistics: [FiLM nets (4L) | conditioner stacks (4L)]
        self.bns = [m[1] for m in spec.film_modules()] + spec.flow_bns()
        nb = len(self.bns)
        self.rm = torch.stack([b.running_mean.detach().to(dev) for b in self.bns])
        self.rv = torch.stack([b.running_var.detach().to(dev) for b in self.bns])
        self.nbt = torch.stack([b.num_batches_tracked.detach().to(dev) for b in self.bns])
        for i, b in enumerate(self.bns):
            b.running_mean.data = self.rm[i]
            b.running_var.data = self.rv[i]
            b.num_batches_tracked.data = self.nbt[i]
        self.nfilm = nb // 2
        self.token = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        # a gradient was written into flat_g since the last zero_grad(set_to_none=True) -- what "p.grad is not None" means
        # for parameters whose .grad views stay attached (networks.optimizers.Adam skips a store without one)
        self.grad_written = had_grad

    def attached(self):
        """The aliasing survives in-place updates, load_state_dict and optimizer steps; module.to()/.cuda()/.float()
        re-assign `.data` and break it (the decoder then builds a new store)."""
        a, b, pv, bn = self.params[0], self.params[-1], self.pviews, self.bns[-1]
        return (a.data_ptr() == pv[0].data_ptr() and b.data_ptr() == pv[-1].data_ptr() and
                bn.running_var.data_ptr() == self.rv[-1].data_ptr() and a.device == self.flat_p.device)

    def rebase_grads(self, buf):
        """Move the gradient buffer onto `buf` (a contiguous fp32 slice of a bigger buffer, same length): contents carried
        over, every parameter's .grad re-pointed.  distributed.GradArena uses this to make the gradients of a whole model
        -- this store, the prior flow's store, every other parameter -- ONE flat message."""
        assert buf.numel() == self.flat_g.numel() and buf.dtype == torch.float32 and buf.is_contiguous() and buf.device == self.flat_g.device
        with torch.no_grad():
            buf.copy_(self.flat_g)
        self.flat_g = buf
        offs, shapes = self._offs, self._shapes
        self.gblocks = [buf[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(6)]
        self.gviews = [buf[o:o + n].view(t.shape) for (o, n), t in zip(self._ranges, self.params)]
        for t, gv in zip(self.params, self.gviews):
            t.grad = gv

    def update_running_fused(self, L, film_mean, film_uvar, stats, momentum):
        """All 8 L running-statistics updates in one launch (r03: ~9 tensor-op launches before); same arithmetic, same bits."""
        check(lib().dpf_flow_train_update_running(L, float(momentum), film_mean.data_ptr(), film_uvar.data_ptr(), stats.data_ptr(),
                                                  self.rm.data_ptr(), self.rv.data_ptr(), self.nbt.data_ptr(), current_stream()),
              "flow_train_update_running")

    def update_running(self, film_mean, film_uvar, flow_mean, flow_uvar, momentum):
        n = self.nfilm
        self.rm.mul_(1.0 - momentum)
        self.rv.mul_(1.0 - momentum)
        self.rm[:n].add_(film_mean, alpha=momentum)
        self.rm[n:].add_(flow_mean, alpha=momentum)
        self.rv[:n].add_(film_uvar, alpha=momentum)
        self.rv[n:].add_(flow_uvar, alpha=momentum)
        self.nbt.add_(1)

    def zero_grad(self):
        self.flat_g.zero_()
        self.grad_written = False
        self.attach_grads(zeroed=True)

    def attach_grads(self, zeroed=False, full=False):
        """Make every parameter's .grad the view of flat_g again: after optimizer.zero_grad(set_to_none=True) the views
        come back zeroed; a .grad that was replaced by another tensor is copied in.  The per-step check looks at
        three sentinel parameters only (first, middle, last); full=True examines all of them."""
        ps, gv = self.params, self.gviews
        if not full and ps[0].grad is gv[0] and ps[-1].grad is gv[-1] and ps[len(ps) // 2].grad is gv[len(ps) // 2]:
            return
        if not zeroed:
            if all(t.grad is None for t in ps):
                self.flat_g.zero_()
                self.grad_written = False
            else:
                self.grad_written = True
                for t, v in zip(ps, gv):
                    if t.grad is None:
                        v.zero_()
                    elif t.grad is not v:
                        v.copy_(t.grad)
        for t, v in zip(ps, gv):
            t.grad = v

    def film_grad_blocks(self):
        """The five FiLM-net gradient blocks of flat_g for the fused backward kernel to ADD into (views attached first)."""
        self.attach_grads()
        gb = self.gblocks
        return gb[1], gb[2], gb[3], gb[4], gb[5]

    def accumulate(self, dcanon, dW0, dgam, dbet, dW1, db1):
        self.attach_grads()
        self.grad_written = True
        gb = self.gblocks
        gb[0].add_(dcanon)
        if dW0 is None:                       # the fused FiLM backward has added its five blocks in place
            return
        gb[1].add_(dW0)
        gb[2].add_(dgam.unsqueeze(1))
        gb[3].add_(dbet.unsqueeze(1))
        gb[4].add_(dW1)
        gb[5].add_(db1.unsqueeze(1))


class _FlowStackTrainFlat(torch.autograd.Function):
    """The stack over a FlatStore: autograd sees p, g and a token; parameter gradients go straight to flat_g."""

    @staticmethod
    def forward(ctx, p, g, token, spec, mode, prec):
        fs = spec.flat
        outs, saved = _forward_core(p.contiguous(), g.contiguous(), spec, mode, prec, *fs.blocks, flat=fs)
        ctx.save_for_backward(*saved)
        ctx.spec, ctx.prec, ctx.mode = spec, prec, mode
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        spec = ctx.spec
        chain, dg, *dparams = _backward_core(spec, ctx.mode, ctx.prec, ctx.saved_tensors, grads, ctx.needs_input_grad[1],
                                             into_flat=True)
        spec.flat.accumulate(*dparams)
        return (chain if ctx.needs_input_grad[0] else None, dg, None, None, None, None)


def run_training_stack(spec, p, g, mode, precision=None):
    """Training-mode forward of the layers of `spec` (DIRECT order) on the HIP path.  Returns
    (ps, mus, lvs): three lists of L (B,3,N) tensors in DIRECT order (views of three (L,B,3,N) blocks),
    attached to autograd; updates the BatchNorm running statistics as nn.BatchNorm1d would."""
    if not p.is_cuda or not g.is_cuda:
        raise RuntimeError("the HIP training path runs on MI355X only (p and g must be CUDA tensors)")
    if p.dtype != torch.float32 or g.dtype != torch.float32:
        raise RuntimeError("p and g must be float32")
    if p.dim() != 3 or p.shape[1] != 3 or g.dim() != 2 or g.shape[0] != p.shape[0]:
        raise RuntimeError("expected p (B,3,N) and g (B,G)")
    if spec.layers[0].f_n_features != F:
        raise RuntimeError("dpf_hip flow kernels are built for f_n_features == 64")
    if g.shape[1] != spec.G:
        raise RuntimeError("g has %d features, the layers expect %d" % (g.shape[1], spec.G))
    if p.shape[0] * p.shape[2] < 2:
        raise ValueError("Expected more than 1 value per channel when training")
    precision = precision or TRAIN_PRECISION
    if precision not in ("bf16x3", "bf16x6", "f16x3"):
        raise ValueError("training precision must be f16x3, bf16x6 or bf16x3")
    with torch.cuda.device(p.device):
        if spec.flat is not None:
            if not spec.flat.attached():
                spec.flat = FlatStore(spec, p.device)            # .to()/.cuda() re-assigned the parameters' data
            if precision == "f16x3":
                precision = spec.f16_range_monitor(spec.flat.blocks[0], p.shape[0] * p.shape[2])
            outs = _FlowStackTrainFlat.apply(p, g, spec.flat.token, spec, mode, PREC[precision])
        else:
            if precision == "f16x3":
                precision = spec.f16_range_monitor(None, p.shape[0] * p.shape[2])
            outs = _FlowStackTrain.apply(p, g, spec, mode, PREC[precision], *spec.all_params())
    L = spec.L
    lvs = list(outs[2 * L:3 * L])
    token = object()                       # losses.total_logvar recognises the whole list and takes the layer-sum
    for i, v in enumerate(lvs):
        v._dpf_pos = (token, i)
    lvs[-1]._dpf_total = (token, L, outs[3 * L])
    return list(outs[:L]), list(outs[L:2 * L]), lvs
