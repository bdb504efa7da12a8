"""RealNVPFlow / RealNVPFlowCouple / GlobalRNVPDecoder -- the latent prior flow on (B, G) codes -- with the
reference's constructor signatures, sub-module, parameter and buffer names (lib/networks/flows.py:163-243,
lib/networks/decoders.py:7-38), so reference checkpoints load unchanged.

forward(g, mode):
  * eval mode, CUDA tensors, no autograd (generation / evaluation): the whole stack in ONE HIP launch
    (csrc/gprior.hip through dpf_gprior_forward); the lists come back as views of three (S,B,G) buffers;
  * training mode on CUDA tensors (GlobalRNVPDecoder): BatchNorm on the statistics of the B rows and the whole
    backward through csrc/gprior_train.hip -- one autograd node, 4 launches per step forward and 5 backward instead
    of ~75 tensor-op launches; DPF_TRAIN_IMPL=torch selects the tensor-op restatement (`forward_torch`), which is
    also what CPU tensors, eval mode under autograd and single RealNVPFlow / RealNVPFlowCouple modules get.
The kernels know RealNVPFlowCouple's two index patterns (even/odd, halves) on an even G; a RealNVPFlow with
other warp_inds runs as tensor ops."""
import ctypes
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .layers import Swish
from .flowlist import FlowList
from .._lib import lib, check, current_stream, MODE


def _needs_autograd(*tensors):
    from .flows import _needs_autograd as point_flow_rule          # (one rule, one warning class: flows.EvalModeAutogradWarning)
    return point_flow_rule(*tensors)


def pattern_code(warp_inds, G):
    """0 even / 1 odd / 2 first half / 3 second half (the kernel's step codes), None for anything else."""
    if G % 2:
        return None
    w = [int(i) for i in warp_inds]
    K = G // 2
    for code, ref in enumerate((range(0, G, 2), range(1, G, 2), range(0, K), range(K, G))):
        if w == list(ref):
            return code
    return None


class _PackedWeights:
    """Mixin: the packed eval-mode weights are rebuilt after anything that may have changed the parameters behind the
    version counters' back -- optimizers update through `.data` -- i.e. on every train()/eval() switch, load_state_dict,
    .to()/.cuda()/.float(); in between, the stack's sentinels catch ordinary in-place edits."""

    def invalidate_packed(self):
        st = self.__dict__.get("_stack")
        if st is not None:
            st.invalidate()

    def train(self, mode=True):
        if mode != self.training:
            self.invalidate_packed()
        return super().train(mode)

    def _apply(self, fn, *a, **kw):
        self.invalidate_packed()
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **kw)


class RealNVPFlow(_PackedWeights, nn.Module):
    def __init__(self, n_features, g_n_features, weight_std=0.01, warp_inds=[0], eps=1e-6):
        super().__init__()
        self.n_features = n_features
        self.g_n_features = g_n_features
        self.weight_std = weight_std
        self.warp_inds = warp_inds
        self.keep_inds = [i for i in range(g_n_features) if i not in set(int(w) for w in warp_inds)]
        self.register_buffer("eps", torch.from_numpy(np.array([eps], dtype=np.float32)))
        for br in ("mu", "logvar"):
            net = nn.Sequential(OrderedDict([
                (br + "_mlp0", nn.Linear(len(self.keep_inds), n_features, bias=False)),
                (br + "_mlp0_bn", nn.BatchNorm1d(n_features)),
                (br + "_mlp0_swish", Swish()),
                (br + "_mlp1", nn.Linear(n_features, len(self.warp_inds), bias=True)),
            ]))
            with torch.no_grad():
                net[-1].weight.normal_(std=weight_std)
                net[-1].bias.zero_()
            setattr(self, "T_%s_0" % br, net)
        self._stack = None

    def canon_pieces(self):
        """The step's tensors in dpf_gprior_pack's canonical order (include/dpf_hip.h)."""
        out = []
        for br in ("mu", "logvar"):
            net = getattr(self, "T_%s_0" % br)
            out += [net[0].weight, net[1].weight, net[1].bias, net[1].running_mean, net[1].running_var, net[3].weight, net[3].bias]
        return [t.detach().reshape(-1) for t in out]

    def forward_torch(self, g, mode="direct"):                  # flows.py:198-213
        gk = g[:, self.keep_inds].contiguous()
        logvar = torch.zeros_like(g)
        mu = torch.zeros_like(g)
        logvar[:, self.warp_inds] = torch.log(self.eps + torch.exp(self.T_logvar_0(gk)))
        mu[:, self.warp_inds] = self.T_mu_0(gk)
        if mode == "direct":
            out = torch.exp(0.5 * logvar) * g + mu
        elif mode == "inverse":
            out = torch.exp(-0.5 * logvar) * (g - mu)
        else:
            raise ValueError(mode)
        return out, mu, logvar

    def forward(self, g, mode="direct"):
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        if _fusable(self, [self], g):
            if self._stack is None:
                self.__dict__["_stack"] = GPriorStack([self])
            _, _, gs, mus, lvs = self._stack.run(g, mode)
            return gs[0], mus[0], lvs[0]
        return self.forward_torch(g, mode)


class RealNVPFlowCouple(_PackedWeights, nn.Module):
    def __init__(self, n_features, g_n_features, weight_std=0.01, pattern=0):
        super().__init__()
        self.n_features = n_features
        self.g_n_features = g_n_features
        self.weight_std = weight_std
        self.pattern = pattern
        idx = np.arange(g_n_features)
        if pattern == 0:
            w1, w2 = idx[::2], idx[1::2]
        elif pattern == 1:
            w1, w2 = idx[:g_n_features // 2], idx[g_n_features // 2:]
        else:
            return                                               # as the reference: no sub-modules for other patterns
        self.nvp1 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=list(w1))
        self.nvp2 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=list(w2))

    def layers(self):
        return [self.nvp1, self.nvp2]

    def forward_torch(self, g, mode="direct"):                  # flows.py:235-243
        if mode == "direct":
            g1, mu1, lv1 = self.nvp1.forward_torch(g, mode)
            g2, mu2, lv2 = self.nvp2.forward_torch(g1, mode)
        elif mode == "inverse":
            g2, mu2, lv2 = self.nvp2.forward_torch(g, mode)
            g1, mu1, lv1 = self.nvp1.forward_torch(g2, mode)
        else:
            raise ValueError(mode)
        return [g1, g2], [mu1, mu2], [lv1, lv2]

    def forward(self, g, mode="direct"):
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        if _fusable(self, self.layers(), g):
            if self.__dict__.get("_stack") is None:
                self.__dict__["_stack"] = GPriorStack(self.layers())
            _, _, gs, mus, lvs = self._stack.run(g, mode)
            return list(gs.unbind(0)), list(mus.unbind(0)), list(lvs.unbind(0))
        return self.forward_torch(g, mode)


class GlobalRNVPDecoder(_PackedWeights, nn.Module):
    def __init__(self, n_flows, n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows = n_flows
        self.n_features = n_features
        self.g_n_features = g_n_features
        self.weight_std = weight_std
        self.flows = nn.ModuleList([RealNVPFlowCouple(n_features, g_n_features, weight_std=weight_std, pattern=(i % 2))
                                    for i in range(n_flows)])

    def coupling_layers(self):
        """The 2*n_flows RealNVPFlow steps in DIRECT order."""
        out = []
        for f in self.flows:
            out += f.layers()
        return out

    def stack(self):
        if self.__dict__.get("_stack") is None:
            self.__dict__["_stack"] = GPriorStack(self.coupling_layers())
        return self._stack

    def flatten_parameters(self):
        """Opt in to the flat parameter store (see PriorFlatStore); returns it.  Call after .cuda()."""
        _adopt_flatstore_methods()
        steps = self.coupling_layers()
        dev = next(self.parameters()).device
        store = self.__dict__.get("_flat")
        if store is None or not store.attached():
            store = self.__dict__["_flat"] = PriorFlatStore(steps, dev)
        return store

    def flat_store(self):
        store = self.__dict__.get("_flat")
        return store if store is not None and store.attached() else None

    def forward_torch(self, g, mode="direct"):                  # decoders.py:21-38
        gs, mus, lvs = [], [], []
        for i in range(self.n_flows):
            if mode == "direct":
                buf = self.flows[i].forward_torch(g if i == 0 else gs[-1], mode)
                gs, mus, lvs = gs + buf[0], mus + buf[1], lvs + buf[2]
            elif mode == "inverse":
                buf = self.flows[-(i + 1)].forward_torch(g if i == 0 else gs[0], mode)
                gs, mus, lvs = buf[0] + gs, buf[1] + mus, buf[2] + lvs
            else:
                raise ValueError(mode)
        return gs, mus, lvs

    def forward(self, g, mode="direct"):
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        steps = self.__dict__.get("_steps") or self.__dict__.setdefault("_steps", self.coupling_layers())
        if self.n_flows and _fusable(self, steps, g):
            _, sum_lv, gs, mus, lvs = self.stack().run(g, mode)
            return FlowList(gs), FlowList(mus), FlowList(lvs, sum_lv)
        from .flows import TRAIN_IMPL
        if self.n_flows and self.training and g.is_cuda and TRAIN_IMPL == "hip" and g.dtype == torch.float32 and \
                self.__dict__.get("_patterns_ok", _patterns_ok(self, steps)):
            return run_training_prior(self, steps, g, mode)
        return self.forward_torch(g, mode)


def _patterns_ok(module, layers):
    ok = module.__dict__.get("_patterns_ok")
    if ok is None:                                                 # warp_inds are fixed at construction
        G = layers[0].g_n_features
        ok = module.__dict__["_patterns_ok"] = all(
            l.g_n_features == G and l.n_features == layers[0].n_features and pattern_code(l.warp_inds, G) is not None for l in layers)
    return ok


def _fusable(module, layers, g):
    """Eval mode on a CUDA tensor without autograd, every step one of the kernel's index patterns."""
    if module.training or not g.is_cuda or _needs_autograd(g):      # as the point decoder: autograd follows the INPUT
        return False
    return _patterns_ok(module, layers)


def _step_params(layers):
    """Every step's tensors in the canonical order; (parameters, their slots in the canon block, the BatchNorm modules)."""
    params, slots, bns, off = [], [], [], 0
    for l in layers:
        for br in ("mu", "logvar"):
            net = getattr(l, "T_%s_0" % br)
            for t in (net[0].weight, net[1].weight, net[1].bias):
                params.append(t); slots.append((off, t.numel())); off += t.numel()
            off += 2 * net[1].num_features                             # running_mean | running_var: not parameters
            for t in (net[3].weight, net[3].bias):
                params.append(t); slots.append((off, t.numel())); off += t.numel()
            bns.append(net[1])
    return params, slots, bns, off


def _train_forward(g, canon, params_only, mode, codes, dims, bn_eps, eps):
    S, G, nf = dims
    B, dev, L_ = g.shape[0], g.device, lib()
    gs, mus, lvs = (torch.empty((S, B, G), dtype=torch.float32, device=dev) for _ in range(3))
    save_h = torch.empty((S, B, 2 * nf), dtype=torch.float32, device=dev)
    stats = torch.empty((S, 2, 2 * nf), dtype=torch.float32, device=dev)
    ws = torch.empty(L_.dpf_gprior_train_workspace_floats(B, G, nf), dtype=torch.float32, device=dev)
    check(L_.dpf_gprior_train_forward(S, B, G, nf, MODE[mode], codes, params_only, canon.data_ptr(), g.data_ptr(), gs.data_ptr(),
                                      mus.data_ptr(), lvs.data_ptr(), save_h.data_ptr(), stats.data_ptr(), ws.data_ptr(), bn_eps, eps,
                                      current_stream()), "gprior_train_forward")
    return gs, mus, lvs, save_h, stats


def _train_backward(g, canon, params_only, gs, mus, lvs, save_h, stats, dlists, mode, codes, dims, bn_eps, eps):
    S, G, nf = dims
    B, dev, L_ = g.shape[0], g.device, lib()
    dg, dcanon = torch.empty_like(g), torch.empty_like(canon)
    ws = torch.empty(L_.dpf_gprior_train_workspace_floats(B, G, nf), dtype=torch.float32, device=dev)
    cg = [t.contiguous() if t is not None else None for t in dlists]                   # alive across the call
    with torch.cuda.device(dev):
        check(L_.dpf_gprior_train_backward(S, B, G, nf, MODE[mode], codes, params_only, canon.data_ptr(), g.data_ptr(), gs.data_ptr(),
                                           mus.data_ptr(), lvs.data_ptr(), save_h.data_ptr(), stats.data_ptr(),
                                           *[t.data_ptr() if t is not None else None for t in cg],
                                           dg.data_ptr(), dcanon.data_ptr(), ws.data_ptr(), bn_eps, eps, current_stream()),
              "gprior_train_backward")
    return dg, dcanon


class PriorFlatStore:
    """All parameters of a GlobalRNVPDecoder in ONE buffer laid out exactly as the training kernels read them (the
    parameters-only canonical block of include/dpf_hip.h) and their gradients in its twin -- the counterpart of
    train_engine.FlatStore for the latent prior flow.  Every nn.Parameter keeps its identity, name and shape; its `.data`
    becomes a view of `flat_p`, its `.grad` a view of `flat_g`.  The forward then needs no gather, the backward adds its
    gradient block with one op instead of feeding 10 tensors per step to AccumulateGrad nodes, networks.optimizers.Adam
    updates the whole store with one op sequence, and the data-parallel exchange is one all-reduce
    (distributed.allreduce_flat_gradients).  As with FlatStore, per-parameter autograd hooks do not fire."""

    def __init__(self, layers, dev):
        params, slots, off = [], [], 0
        for l in layers:
            for br in ("mu", "logvar"):
                net = getattr(l, "T_%s_0" % br)
                for t in (net[0].weight, net[1].weight, net[1].bias, net[3].weight, net[3].bias):
                    params.append(t); slots.append((off, t.numel())); off += t.numel()
        self.params, self.total, self._slots = params, off, slots
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        with torch.no_grad():
            torch._foreach_copy_([self.flat_p[o:o + n] for o, n in slots], [t.detach().reshape(-1).to(dev) for t in params])
        self.pviews = [self.flat_p[o:o + n].view(t.shape) for (o, n), t in zip(slots, params)]
        self.gviews = [self.flat_g[o:o + n].view(t.shape) for (o, n), t in zip(slots, params)]
        self.grad_written = any(t.grad is not None for t in params)     # see FlatStore.grad_written
        for t, pv, gv in zip(params, self.pviews, self.gviews):
            if t.grad is not None:
                gv.copy_(t.grad)
            t.data = pv
            t.grad = gv
            t._dpf_flat = self
        from .train_engine import hook_grad_written
        hook_grad_written(params)
        self.token = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)

    def attached(self):
        a, b = self.params[0], self.params[-1]
        return a.data_ptr() == self.pviews[0].data_ptr() and b.data_ptr() == self.pviews[-1].data_ptr() and a.device == self.flat_p.device

    def rebase_grads(self, buf):
        """As FlatStore.rebase_grads: the gradient buffer becomes `buf` (a slice of distributed.GradArena's message)."""
        assert buf.numel() == self.flat_g.numel() and buf.dtype == torch.float32 and buf.is_contiguous() and buf.device == self.flat_g.device
        with torch.no_grad():
            buf.copy_(self.flat_g)
        self.flat_g = buf
        self.gviews = [buf[o:o + n].view(t.shape) for (o, n), t in zip(self._slots, self.params)]
        for t, gv in zip(self.params, self.gviews):
            t.grad = gv

    def accumulate(self, dcanon):
        self.attach_grads()
        self.grad_written = True
        self.flat_g.add_(dcanon)


def _adopt_flatstore_methods():
    from .train_engine import FlatStore
    PriorFlatStore.attach_grads = FlatStore.attach_grads          # same bookkeeping over params / gviews / flat_g
    PriorFlatStore.zero_grad = FlatStore.zero_grad


class _GPriorTrainFlat(torch.autograd.Function):
    """The training-mode stack over a PriorFlatStore: autograd sees g and a token; parameter gradients go straight to flat_g."""

    @staticmethod
    def forward(ctx, g, token, store, mode, codes, dims, bn_eps, eps):
        g = g.contiguous()
        gs, mus, lvs, save_h, stats = _train_forward(g, store.flat_p, 1, mode, codes, dims, bn_eps, eps)
        ctx.save_for_backward(g, gs, mus, lvs, save_h, stats)
        ctx.cfg = (store, mode, codes, dims, bn_eps, eps, store.flat_p._version)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(stats)
        return gs, mus, lvs, stats

    @staticmethod
    def backward(ctx, d_gs, d_mus, d_lvs, _):
        g, gs, mus, lvs, save_h, stats = ctx.saved_tensors
        store, mode, codes, dims, bn_eps, eps, version = ctx.cfg
        if store.flat_p._version != version:
            raise RuntimeError("the flattened parameters were modified in place between forward and backward")
        dg, dcanon = _train_backward(g, store.flat_p, 1, gs, mus, lvs, save_h, stats, (d_gs, d_mus, d_lvs), mode, codes, dims, bn_eps, eps)
        store.accumulate(dcanon)
        return (dg if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None)


class _GPriorTrain(torch.autograd.Function):
    """The whole training-mode stack as one node: inputs g and every parameter, outputs the three (S,B,G) blocks."""

    @staticmethod
    def forward(ctx, g, mode, codes, dims, bn_eps, eps, slots, total, *params):
        S, G, nf = dims
        B = g.shape[0]
        L_ = lib()
        g = g.contiguous()
        dev = g.device
        canon = torch.zeros(total, dtype=torch.float32, device=dev)
        torch._foreach_copy_([canon[o:o + n] for o, n in slots], [p.detach().reshape(-1) for p in params])
        gs, mus, lvs, save_h, stats = _train_forward(g, canon, 0, mode, codes, dims, bn_eps, eps)
        ctx.save_for_backward(g, canon, gs, mus, lvs, save_h, stats, *params)
        ctx.cfg = (mode, codes, dims, bn_eps, eps, slots)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(stats)
        return gs, mus, lvs, stats

    @staticmethod
    def backward(ctx, d_gs, d_mus, d_lvs, _):
        g, canon, gs, mus, lvs, save_h, stats = ctx.saved_tensors[:7]
        params = ctx.saved_tensors[7:]
        mode, codes, (S, G, nf), bn_eps, eps, slots = ctx.cfg
        B = g.shape[0]
        L_ = lib()
        dev = g.device
        dg, dcanon = _train_backward(g, canon, 0, gs, mus, lvs, save_h, stats, (d_gs, d_mus, d_lvs), mode, codes, (S, G, nf), bn_eps, eps)
        grads = [torch.empty_like(p) for p in params]
        torch._foreach_copy_(grads, [dcanon[o:o + n].view_as(p) for (o, n), p in zip(slots, params)])
        return (dg if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None, *grads)


def run_training_prior(module, layers, g, mode):
    """Training-mode forward of GlobalRNVPDecoder on the HIP path: three python lists of (B,G) tensors in DIRECT order
    (views of three blocks, attached to autograd); updates the BatchNorm running statistics as nn.BatchNorm1d does."""
    if g.dim() != 2 or g.shape[1] != layers[0].g_n_features:
        raise RuntimeError("expected g (B,%d)" % layers[0].g_n_features)
    if g.shape[0] < 2:
        raise ValueError("Expected more than 1 value per channel when training")      # as nn.BatchNorm1d
    cache = module.__dict__.get("_train_plan")
    if cache is None:
        params, slots, bns, total = _step_params(layers)
        S, G, nf = len(layers), layers[0].g_n_features, layers[0].n_features
        codes = (ctypes.c_int * S)(*[pattern_code(l.warp_inds, G) for l in layers])
        assert total == S * lib().dpf_gprior_canon_floats(G, nf)
        cache = module.__dict__["_train_plan"] = (params, slots, bns, total, codes, (S, G, nf), float(layers[0].eps.item()))
    params, slots, bns, total, codes, dims, eps = cache
    store = module.__dict__.get("_flat")
    if store is not None and not store.attached():               # .to()/.cuda()/.float() re-assigned the parameters' data
        store = module.__dict__["_flat"] = PriorFlatStore(layers, g.device)
    with torch.cuda.device(g.device):
        if store is not None:
            gs, mus, lvs, stats = _GPriorTrainFlat.apply(g, store.token, store, mode, codes, dims, bns[0].eps, eps)
        else:
            gs, mus, lvs, stats = _GPriorTrain.apply(g, mode, codes, dims, bns[0].eps, eps, slots, total, *params)
        # running statistics (nn.BatchNorm1d: momentum 0.1, unbiased variance), multi-tensor
        B, nf = g.shape[0], dims[2]
        with torch.no_grad():
            m = bns[0].momentum
            means = list(stats[:, 0].reshape(-1, nf).unbind(0))
            uvars = list((stats[:, 1] * (B / (B - 1.0))).reshape(-1, nf).unbind(0))
            rms, rvs = [b.running_mean for b in bns], [b.running_var for b in bns]
            torch._foreach_mul_(rms, 1.0 - m)
            torch._foreach_add_(rms, means, alpha=m)
            torch._foreach_mul_(rvs, 1.0 - m)
            torch._foreach_add_(rvs, uvars, alpha=m)
            torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
    lv_list = list(lvs.unbind(0))
    # the layer-sum of the log-variances, which is all GaussianFlowNLL wants of them (losses.py:22): one reduction over the
    # (S,B,G) block and one expand in the backward instead of S - 1 adds and S gradient slices stacked back; tagged like the
    # point decoder's list (networks.losses.total_logvar recognises the whole list)
    token = object()
    for i, v in enumerate(lv_list):
        v._dpf_pos = (token, i)
    lv_list[-1]._dpf_total = (token, len(lv_list), lvs.sum(0))
    return list(gs.unbind(0)), list(mus.unbind(0)), lv_list


class GPriorStack:
    """Packed weights of an ordered list of RealNVPFlow steps (built by dpf_gprior_pack once per weight version)
    and the launch.  torch is used for device buffers and the stream handle only."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.G, self.nf = self.layers[0].g_n_features, self.layers[0].n_features
        self.codes = [pattern_code(l.warp_inds, self.G) for l in self.layers]
        self._packed = None
        self._state = None
        self._sentinels = []
        for l in (self.layers[0], self.layers[-1]):
            self._sentinels += [l.T_mu_0[0].weight, l.T_mu_0[1].running_var, l.T_logvar_0[3].weight, l.T_logvar_0[1].running_mean]

    def invalidate(self):
        self._packed = None

    def _ensure(self, device):
        state = tuple((t._version, t.data_ptr()) for t in self._sentinels)
        if state != self._state or self._packed is None or self._packed.device != device:
            L_ = lib()
            S = len(self.layers)
            pieces = []
            for l in self.layers:
                pieces += l.canon_pieces()
            canon = torch.cat(pieces).to(device=device, dtype=torch.float32).contiguous()
            assert canon.numel() == S * L_.dpf_gprior_canon_floats(self.G, self.nf), (canon.numel(), S, self.G, self.nf)
            packed = torch.empty(L_.dpf_gprior_packed_floats(S, self.G, self.nf), dtype=torch.float32, device=device)
            bn = self.layers[0].T_mu_0[1]
            check(L_.dpf_gprior_pack(S, self.G, self.nf, bn.eps, canon.data_ptr(), packed.data_ptr(), current_stream()), "gprior_pack")
            self._packed, self._state = packed, state
            self._eps = float(self.layers[0].eps.item())
        return self._packed

    def run(self, g, mode, want_lists=True):
        """g (B,G) fp32 CUDA -> (g_out (B,G), sum_logvar (B,G), gs, mus, lvs (S,B,G) in DIRECT order or None)."""
        import ctypes
        if not g.is_cuda:
            raise RuntimeError("the fused prior flow runs on MI355X only (g must be a CUDA tensor); there is no CPU fallback")
        if g.dtype != torch.float32 or g.dim() != 2 or g.shape[1] != self.G:
            raise RuntimeError("expected g (B,%d) float32" % self.G)
        g = g.contiguous()
        B, S = g.shape[0], len(self.layers)
        with torch.cuda.device(g.device):
            packed = self._ensure(g.device)
            g_out, sum_lv = torch.empty_like(g), torch.empty_like(g)
            gs, mus, lvs = (torch.empty((S, B, self.G), dtype=torch.float32, device=g.device) for _ in range(3)) if want_lists \
                else (None, None, None)
            codes = (ctypes.c_int * S)(*self.codes)
            ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
            check(lib().dpf_gprior_forward(S, B, self.G, self.nf, MODE[mode], codes, packed.data_ptr(), g.data_ptr(), ptr(gs),
                                           ptr(mus), ptr(lvs), sum_lv.data_ptr(), g_out.data_ptr(), self._eps, current_stream()),
                  "gprior_forward")
        return g_out, sum_lv, gs, mus, lvs
