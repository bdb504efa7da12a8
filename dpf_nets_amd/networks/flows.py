"""CondRealNVPFlow3D / CondRealNVPFlow3DTriple with the reference's constructor
signatures, sub-module and parameter names (lib/networks/flows.py:10-160), so
reference checkpoints load unchanged.

forward(p, g, mode) -> (p_out, mu, logvar), p (B,3,N), g (B,G):
  * eval mode (BatchNorm frozen, the per-point map of evaluate()/inference):
    the fused HIP stack (csrc/flow.hip) -- no torch ops on the path;
  * training mode (BatchNorm batch statistics + autograd) on CUDA tensors: the HIP
    training kernels (csrc/flow_train.hip via networks/train_engine.py), forward and
    backward; DPF_TRAIN_IMPL=torch selects the tensor-op restatement of
    flows.py:95-117 instead (`forward_torch`, also what CPU tensors get: the
    reference-shaped module the CPU tests check against the golden vectors).
"""
import os

from collections import OrderedDict

import torch
import torch.nn as nn

from .layers import SharedDot, Swish
from .engine import FlowStack

TRAIN_IMPL = os.environ.get("DPF_TRAIN_IMPL", "hip")


def use_hip_training(module, p):
    return module.training and p.is_cuda and TRAIN_IMPL == "hip"


TRAIN_FLAT = os.environ.get("DPF_TRAIN_FLAT", "0") == "1"


def stack_spec(owner, layers):
    """The static layout description of `layers`, cached on `owner` per number of layers."""
    from .train_engine import StackSpec
    cache = owner.__dict__.setdefault("_train_specs", {})
    spec = cache.get(len(layers))
    if spec is None:
        spec = cache[len(layers)] = StackSpec(layers)
    return spec


def train_stack(owner, layers, p, g, mode, allow_flat=False):
    """Run `layers` (DIRECT order) through the HIP training path.  allow_flat: DPF_TRAIN_FLAT=1 may move the
    parameters into a flat store on first use (the decoder passes it for its full stack only)."""
    from .train_engine import run_training_stack
    spec = stack_spec(owner, layers)
    if allow_flat and TRAIN_FLAT and spec.flat is None:
        spec.flatten(p.device)
    return run_training_stack(spec, p, g, mode)


class EvalModeAutogradWarning(RuntimeWarning):
    """An eval()-mode module was called under autograd: the call is served by PyTorch tensor operations, not by the HIP kernels."""


def _needs_autograd(*tensors):
    """True when an eval()-mode call must stay differentiable with respect to its INPUTS (the reference's
    CondRealNVPFlow3D.forward is: flows.py:95-117).  The fused HIP stacks have a backward pass for training-mode BatchNorm only
    (csrc/flow_train.hip); an eval-mode call under autograd therefore runs the reference's op sequence on ATen (~30 kernels per
    layer) -- correct, differentiable, and some 50x slower than the fused stack.  That is a SECOND backend behind forward(), so it
    is loud: one EvalModeAutogradWarning per call site.  The reference's own loops never take it
    (evaluating.py:58-59 runs under no_grad; training.py:37-56 in train() mode)."""
    need = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
    if need:
        import warnings
        warnings.warn("eval()-mode flow called with inputs that require grad: served by PyTorch tensor operations (differentiable, "
                      "~30 ATen kernels per coupling layer), not by the fused HIP stack; call under torch.no_grad() for the HIP "
                      "path, or in train() mode for the HIP training kernels", EvalModeAutogradWarning, stacklevel=3)
    return need


class CondRealNVPFlow3D(nn.Module):
    def __init__(self, f_n_features, g_n_features, weight_std=0.01, warp_inds=[0],
                 centered_translation=False, eps=1e-6):
        super().__init__()
        self.f_n_features, self.g_n_features = f_n_features, g_n_features
        self.weight_std = weight_std
        self.warp_inds = list(warp_inds)
        self.keep_inds = [c for c in (0, 1, 2) if c not in self.warp_inds]
        self.centered_translation = centered_translation          # stored, never used (flows.py:20)
        self.eps_value = float(eps)
        self.register_buffer("eps", torch.tensor([eps], dtype=torch.float32))
        Fh, G, nk, nw = f_n_features, g_n_features, len(self.keep_inds), len(self.warp_inds)
        for br in ("mu", "logvar"):                                # registration order = state-dict order
            setattr(self, "T_%s_0" % br, nn.Sequential(OrderedDict([
                ("%s_sd0" % br, SharedDot(nk, Fh, 1)),
                ("%s_sd0_bn" % br, nn.BatchNorm1d(Fh)),
                ("%s_sd0_relu" % br, nn.ReLU(inplace=True)),
                ("%s_sd1" % br, SharedDot(Fh, Fh, 1)),
                ("%s_sd1_bn" % br, nn.BatchNorm1d(Fh, affine=False)),
            ])))
            for s in ("w", "b"):
                film = nn.Sequential(OrderedDict([
                    ("%s_sd1_film_%s0" % (br, s), nn.Linear(G, Fh, bias=False)),
                    ("%s_sd1_film_%s0_bn" % (br, s), nn.BatchNorm1d(Fh)),
                    ("%s_sd1_film_%s0_swish" % (br, s), Swish()),
                    ("%s_sd1_film_%s1" % (br, s), nn.Linear(Fh, Fh, bias=True)),
                ]))
                setattr(self, "T_%s_0_cond_%s" % (br, s), film)
            setattr(self, "T_%s_1" % br, nn.Sequential(OrderedDict([
                ("%s_sd1_relu" % br, nn.ReLU(inplace=True)),
                ("%s_sd2" % br, SharedDot(Fh, nw, 1, bias=True)),
            ])))
            with torch.no_grad():                                  # flows.py:52-58 / 87-93
                for last in (getattr(self, "T_%s_0_cond_w" % br)[-1], getattr(self, "T_%s_0_cond_b" % br)[-1],
                             getattr(self, "T_%s_1" % br)[-1]):
                    last.weight.normal_(std=weight_std)
                    last.bias.zero_()
        object.__setattr__(self, "_stack", None)
        self.precision = None                                      # None -> engine.DEFAULT_PRECISION
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate_packed())

    # -- packed-weight cache invalidation --------------------------------------
    def invalidate_packed(self):
        if self._stack is not None:
            self._stack.invalidate()

    def train(self, mode=True):
        if mode != self.training:
            self.invalidate_packed()
        return super().train(mode)

    def _apply(self, fn, *a, **kw):
        self.invalidate_packed()
        return super()._apply(fn, *a, **kw)

    # -- the two paths -----------------------------------------------------------
    def _conditioner(self, br, x, g):
        h = getattr(self, "T_%s_0" % br)(x)
        a = self.eps + torch.exp(getattr(self, "T_%s_0_cond_w" % br)(g).unsqueeze(2))
        return getattr(self, "T_%s_1" % br)(a * h + getattr(self, "T_%s_0_cond_b" % br)(g).unsqueeze(2))

    def forward_torch(self, p, g, mode="direct"):
        """flows.py:95-117 on tensor ops (training-mode BatchNorm, differentiable)."""
        x = p[:, self.keep_inds, :].contiguous()
        logvar = torch.zeros_like(p)
        mu = torch.zeros_like(p)
        logvar[:, self.warp_inds, :] = nn.functional.softsign(self._conditioner("logvar", x, g))
        mu[:, self.warp_inds, :] = self._conditioner("mu", x, g)
        scale = torch.sqrt(self.eps + torch.exp(logvar))
        if mode == "direct":
            p_out = scale * p + mu
        elif mode == "inverse":
            p_out = (p - mu) / scale
        else:
            raise ValueError(mode)
        return p_out, mu, logvar

    def forward(self, p, g, mode="direct"):
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        if use_hip_training(self, p):
            ps, mus, lvs = train_stack(self, [self], p, g, mode)
            return ps[0], mus[0], lvs[0]
        if self.training or _needs_autograd(p, g):
            return self.forward_torch(p, g, mode)
        if self._stack is None:
            object.__setattr__(self, "_stack", FlowStack([self]))
        p_out, _, ps, mus, lvs = self._stack.run(p, g, mode, self.precision, want_lists=True)
        return p_out, mus[0], lvs[0]


class CondRealNVPFlow3DTriple(nn.Module):
    WARPS = {0: ([0], [1], [2]), 1: ([0, 1], [0, 2], [1, 2])}     # flows.py:129-148

    def __init__(self, f_n_features, g_n_features, weight_std=0.02, pattern=0, centered_translation=False):
        super().__init__()
        self.f_n_features, self.g_n_features = f_n_features, g_n_features
        self.weight_std, self.pattern, self.centered_translation = weight_std, pattern, centered_translation
        for i, warp in enumerate(self.WARPS[pattern]):
            setattr(self, "nvp%d" % (i + 1), CondRealNVPFlow3D(
                f_n_features, g_n_features, weight_std=weight_std, warp_inds=list(warp),
                centered_translation=centered_translation))

    def layers(self):
        return [self.nvp1, self.nvp2, self.nvp3]

    def _chain(self, p, g, mode, call):                            # flows.py:151-160
        if mode == "direct":
            p1, mu1, lv1 = call(self.nvp1, p, g, mode)
            p2, mu2, lv2 = call(self.nvp2, p1, g, mode)
            p3, mu3, lv3 = call(self.nvp3, p2, g, mode)
        elif mode == "inverse":
            p3, mu3, lv3 = call(self.nvp3, p, g, mode)
            p2, mu2, lv2 = call(self.nvp2, p3, g, mode)
            p1, mu1, lv1 = call(self.nvp1, p2, g, mode)
        else:
            raise ValueError(mode)
        return [p1, p2, p3], [mu1, mu2, mu3], [lv1, lv2, lv3]

    def forward_torch(self, p, g, mode="direct"):
        return self._chain(p, g, mode, lambda lyr, pp, gg, mm: lyr.forward_torch(pp, gg, mm))

    def forward(self, p, g, mode="direct"):
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        if use_hip_training(self, p):                              # the three layers as one autograd node
            ps, mus, lvs = train_stack(self, self.layers(), p, g, mode)
            return ps, mus, lvs
        return self._chain(p, g, mode, lambda lyr, pp, gg, mm: lyr(pp, gg, mode=mm))
