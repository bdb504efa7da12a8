"""PointNetCloudEncoder with the reference's constructor signature, sub-module and parameter names
(lib/networks/encoders.py:9-28), so reference checkpoints load unchanged.

forward(input (B,3,N)) -> per-point features (B,512,N):
  * eval mode on CUDA tensors, for the architecture every config uses (3 -> 64 -> [128, 256, 512]): the fused HIP
    kernel of csrc/encoder.hip.  The models reduce the features with `torch.max(features, dim=2)[0]`
    (models.py:85,131,175) and use nothing else, so forward returns a `PointFeatures`: that max is answered from
    the kernel's own (B,512) result and the (B,512,N) tensor (134 MB at B=32, N=2048) is only produced -- by the
    same kernel with its feature output switched on -- if something else is asked of it;
  * training mode on CUDA tensors, same architecture: forward returns a
    `TrainPointFeatures`; its max over the points runs csrc/encoder_train.hip (batch-statistics BatchNorm, running
    statistics updated in place, argmax kept) and is an autograd node whose backward is the HIP backward pass to
    the twelve parameter gradients (and to the input when it requires grad).  Any other use of the features
    materialises them with tensor ops;
  * eval mode with a differentiable input, other widths, BatchNorm without running statistics or with momentum=None: the
    tensor-op path (`self.features(input)`), on PyTorch-ROCm;
  * eval mode on CPU tensors raises (no CPU fallback).
"""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from .._lib import lib, check, current_stream, PREC
from .layers import SharedDot

_HIP_ARCH = (3, 64, (128, 256, 512))


class _MaxResult:
    """(values, indices) of torch.max(features, dim=2): the values come from the fused kernel, the indices (which
    the reference never uses) from the materialised features."""

    def __init__(self, feats, keepdim):
        self._f, self._keep = feats, keepdim
        v = feats.max_over_points()
        self.values = v.unsqueeze(2) if keepdim else v

    @property
    def indices(self):
        return torch.max(self._f.tensor(), dim=2, keepdim=self._keep)[1]

    def __getitem__(self, i):
        return (self.values, self.indices)[i] if i in (1, -1) else (self.values,)[i]

    def __iter__(self):
        return iter((self.values, self.indices))

    def __len__(self):
        return 2


class PointFeatures:
    """Lazy (B,512,N) per-point features of one eval-mode encoder call."""

    def __init__(self, encoder, x):
        self._enc, self._x = encoder, x
        self._max = None
        self._full = None
        self.shape = torch.Size((x.shape[0], encoder.n_features[-1], x.shape[2]))
        self.device, self.dtype = x.device, torch.float32

    def _run(self, want_features):
        enc, x = self._enc, self._x
        B, _, N = x.shape
        packed = enc._packed(x.device)
        gmax = torch.empty((B, self.shape[1]), dtype=torch.float32, device=x.device)
        feat = torch.empty(tuple(self.shape), dtype=torch.float32, device=x.device) if want_features else None
        with torch.cuda.device(x.device):
            check(lib().dpf_encoder_forward(B, N, PREC[enc.precision], packed.data_ptr(), x.data_ptr(), gmax.data_ptr(),
                                            feat.data_ptr() if want_features else None, current_stream()), "encoder_forward")
        self._max = gmax
        if want_features:
            self._full = feat

    def max_over_points(self):
        """(B,512): torch.max(features, dim=2)[0]"""
        if self._max is None:
            self._run(False)
        return self._max

    def tensor(self):
        """the (B,512,N) features as a tensor"""
        if self._full is None:
            self._run(True)
        return self._full

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 3

    def max(self, dim=None, keepdim=False):
        return torch.max(self, dim=dim, keepdim=keepdim) if dim is not None else self.max_over_points().max()

    def __getattr__(self, name):          # anything else a tensor can do
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.max, torch.Tensor.max, torch.amax) and len(args) >= 1 and isinstance(args[0], PointFeatures):
            dim = args[1] if len(args) > 1 else kwargs.get("dim")
            keep = args[2] if len(args) > 2 else kwargs.get("keepdim", False)
            if dim in (2, -1) or dim in ((2,), (-1,), [2], [-1]):
                if func is torch.amax:
                    v = args[0].max_over_points()
                    return v.unsqueeze(2) if keep else v
                return _MaxResult(args[0], keep)
        def unwrap(a):
            if isinstance(a, PointFeatures):
                return a.tensor()
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(v) for v in a)
            return a
        return func(*unwrap(args), **{k: unwrap(v) for k, v in kwargs.items()})


def _delegate(name):
    def op(self, *args, **kwargs):
        return getattr(self.tensor(), name)(*args, **kwargs)
    op.__name__ = name
    return op


for _n in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "neg", "matmul", "rmatmul",
           "lt", "le", "gt", "ge", "eq", "ne", "len", "iter", "repr", "array"):
    setattr(PointFeatures, "__%s__" % _n, _delegate("__%s__" % _n))
PointFeatures.__hash__ = object.__hash__


class _TrainPool(torch.autograd.Function):
    """pooled = max over the points of the training-mode encoder (encoders.py:27-28 + models.py:131) on
    dpf_encoder_train_forward / dpf_encoder_train_backward.  params: W, gamma, beta of the four layers."""

    @staticmethod
    def forward(ctx, enc, x, *params):
        B, _, N = x.shape
        dev = x.device
        L_ = lib()
        canon = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in enc._layer_tensors()]).contiguous()
        ws = torch.empty(L_.dpf_encoder_train_workspace_bytes(B, N), dtype=torch.uint8, device=dev)
        pooled = torch.empty((B, enc.n_features[-1]), dtype=torch.float32, device=dev)
        bns = [getattr(enc.features, n + "_bn") for n in _LAYERS]
        ptrs = (ctypes.c_void_p * 8)(*[t.data_ptr() for bn in bns for t in (bn.running_mean, bn.running_var)])
        with torch.cuda.device(dev):
            check(L_.dpf_encoder_train_forward(B, N, PREC[enc.train_precision], canon.data_ptr(), x.data_ptr(), ws.data_ptr(), pooled.data_ptr(), None,
                                               ptrs, float(bns[0].momentum), current_stream()), "encoder_train_forward")
        for bn in bns:                                                    # written behind torch's back
            torch.autograd.graph.increment_version(bn.running_mean)
            torch.autograd.graph.increment_version(bn.running_var)
        torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
        # canon, x and the workspace go through save_for_backward: version counters catch an in-place edit of x between
        # forward and backward, and a second backward (retain_graph=True) finds them again
        ctx.dims = (B, N)
        ctx.save_for_backward(pooled, canon, x, ws)
        ctx.shapes = [p.shape for p in params]
        return pooled

    @staticmethod
    def backward(ctx, g):
        B, N = ctx.dims
        pooled, canon, x, ws = ctx.saved_tensors
        dcanon = torch.empty_like(canon)
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        g32 = g.contiguous().to(torch.float32)                            # a local: it must outlive the launch's enqueue
        with torch.cuda.device(x.device):
            check(lib().dpf_encoder_train_backward(B, N, canon.data_ptr(), x.data_ptr(), ws.data_ptr(), pooled.data_ptr(),
                                                   g32.data_ptr(), dcanon.data_ptr(),
                                                   dx.data_ptr() if dx is not None else None, current_stream()),
                  "encoder_train_backward")
        grads, off = [], 0
        cin = _HIP_ARCH[0]
        for i, cout in enumerate((_HIP_ARCH[1],) + tuple(_HIP_ARCH[2])):
            for n in (cout * cin, cout, cout):                            # W, gamma, beta; then the running-statistics slots
                grads.append(dcanon[off:off + n].view(ctx.shapes[len(grads)]))
                off += n
            off += 2 * cout
            cin = cout
        return (None, dx) + tuple(grads)


_LAYERS = ("init_sd", "sd0", "sd1", "sd2")


class TrainPointFeatures(PointFeatures):
    """Lazy (B,512,N) features of one TRAINING-mode encoder call: the max over the points is the HIP autograd node
    (batch statistics are taken, running statistics updated, exactly once -- by whichever use comes first)."""

    def __init__(self, encoder, x):
        super().__init__(encoder, x)
        self.requires_grad = torch.is_grad_enabled() and any(p.requires_grad for p in encoder.parameters())

    def max_over_points(self):
        if self._max is None:
            enc = self._enc
            if self._full is not None:                                   # the features were asked for first
                self._max = torch.max(self._full, dim=2)[0]
            else:
                params = [t for name in _LAYERS for t in (getattr(enc.features, name).weight, getattr(enc.features, name + "_bn").weight,
                                                          getattr(enc.features, name + "_bn").bias)]
                self._max = _TrainPool.apply(enc, self._x, *params)
        return self._max

    def tensor(self):
        if self._full is None:
            enc = self._enc
            if self._max is None:
                self._full = enc.forward_torch(self._x)
            else:                                                         # statistics already taken by the HIP pass
                bns = [getattr(enc.features, n + "_bn") for n in _LAYERS]
                keep = [(bn.momentum, bn.num_batches_tracked.clone()) for bn in bns]
                try:
                    for bn in bns:
                        bn.momentum = 0.0
                    self._full = enc.forward_torch(self._x)
                finally:
                    for bn, (m, nb) in zip(bns, keep):
                        bn.momentum = m
                        bn.num_batches_tracked.copy_(nb)
        return self._full


class PointNetCloudEncoder(nn.Module):
    def __init__(self, init_n_channels, init_n_features, n_features):
        super().__init__()
        self.init_n_channels, self.init_n_features, self.n_features = init_n_channels, init_n_features, n_features
        self.features = nn.Sequential(OrderedDict([                       # encoders.py:15-25
            ("init_sd", SharedDot(init_n_channels, init_n_features, 1, bias=False)),
            ("init_sd_bn", nn.BatchNorm1d(init_n_features)),
            ("init_sd_relu", nn.ReLU(inplace=True)),
        ]))
        for i in range(len(self.n_features)):
            cur = init_n_features if i == 0 else n_features[i - 1]
            self.features.add_module("sd{}".format(i), SharedDot(cur, n_features[i], 1, bias=False))
            self.features.add_module("sd{}_bn".format(i), nn.BatchNorm1d(n_features[i]))
            self.features.add_module("sd{}_relu".format(i), nn.ReLU(inplace=True))
        self.precision = "bf16x3"            # "bf16x3" (default, ~1e-5), "bf16x6" (fp32-class), "bf16"
        self.hip_training = True             # training mode on csrc/encoder_train.hip (False: tensor ops)
        self.train_precision = "bf16x6"      # forward contractions of the training path ("bf16x6" fp32-class, "bf16x3")
        object.__setattr__(self, "_pack_cache", {})

    def hip_supported(self):
        return (self.init_n_channels, self.init_n_features, tuple(self.n_features)) == _HIP_ARCH and \
            all(getattr(self.features, n).eps == 1e-5 for n in ("init_sd_bn", "sd0_bn", "sd1_bn", "sd2_bn"))

    def _layer_tensors(self):
        out = []
        for name in ("init_sd", "sd0", "sd1", "sd2"):
            sd, bn = getattr(self.features, name), getattr(self.features, name + "_bn")
            out += [sd.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return out

    def _packed(self, dev):
        """bf16 MFMA fragments of the BatchNorm-folded weights, cached per (weight version, precision, device)"""
        ts = self._layer_tensors()
        state = tuple((t._version, t.data_ptr()) for t in ts)
        key = (self.precision, str(dev))
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == state:
            return hit[1]
        L_ = lib()
        with torch.no_grad():
            canon = torch.cat([t.detach().reshape(-1) for t in ts]).to(device=dev, dtype=torch.float32).contiguous()
        assert canon.numel() == L_.dpf_encoder_canon_floats(), canon.numel()
        packed = torch.empty(L_.dpf_encoder_packed_bytes(PREC[self.precision]), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(L_.dpf_encoder_pack(PREC[self.precision], canon.data_ptr(), packed.data_ptr(), current_stream()), "encoder_pack")
        self._pack_cache.clear()
        self._pack_cache[key] = (state, packed)
        return packed

    def forward_torch(self, input):
        return self.features(input)                                       # encoders.py:27-28

    def hip_training_supported(self, input):
        bns = [getattr(self.features, n + "_bn") for n in _LAYERS]
        return self.hip_training and input.is_cuda and input.dtype == torch.float32 and input.dim() == 3 and input.shape[1] == 3 \
            and input.shape[0] * input.shape[2] >= 2 and input.shape[0] <= 65535 \
            and all(bn.track_running_stats and bn.momentum is not None and bn.running_mean is not None
                    and bn.running_mean.dtype == torch.float32 and bn.weight is not None and bn.weight.dtype == torch.float32
                    for bn in bns) \
            and len({bn.momentum for bn in bns}) == 1 and all(getattr(self.features, n).weight.dtype == torch.float32 for n in _LAYERS)

    def forward(self, input):
        # other widths, or a differentiable input in eval mode: tensor ops
        if not self.hip_supported() or (torch.is_grad_enabled() and input.requires_grad and not self.training):
            return self.forward_torch(input)
        if self.training:
            if not self.hip_training_supported(input):
                return self.forward_torch(input)
            return TrainPointFeatures(self, input.contiguous())
        if not input.is_cuda:
            raise RuntimeError("PointNetCloudEncoder: the eval-mode path runs on MI355X only (input must be a CUDA tensor); "
                               "there is no CPU fallback")
        if input.dtype != torch.float32 or input.dim() != 3 or input.shape[1] != 3:
            raise RuntimeError("PointNetCloudEncoder: expected a float32 (B,3,N) tensor")
        return PointFeatures(self, input.contiguous())


class FeatureEncoder(nn.Module):
    """lib/networks/encoders.py:31-83: n_layers x [Linear(no bias) . BatchNorm1d . Swish] on a (B, C) code, then a `mus`
    head and (unless deterministic) a `logvars` head.  O(B) tensor ops on PyTorch-ROCm (SURVEY: stays host code) -- here
    so that the whole autoencoder (networks/models.py) is built from this package and its gradients form ONE flat
    message (distributed.GradArena).  Same sub-module / parameter names and initialisation as the reference's."""

    def __init__(self, n_layers, in_features, latent_space_size, deterministic=False, batch_norm=True, mu_weight_std=0.001,
                 mu_bias=0.0, logvar_weight_std=0.01, logvar_bias=0.0, easy_init=False):
        super().__init__()
        from .layers import Swish
        self.n_layers, self.in_features, self.latent_space_size = n_layers, in_features, latent_space_size
        self.deterministic, self.batch_norm = deterministic, batch_norm
        if n_layers > 0:
            self.features = nn.Sequential()
            for i in range(n_layers):
                self.features.add_module("mlp{}".format(i), nn.Linear(in_features, in_features, bias=False))
                if batch_norm:
                    self.features.add_module("mlp{}_bn".format(i), nn.BatchNorm1d(in_features))
                self.features.add_module("mlp{}_swish".format(i), Swish())
        self.mus = nn.Sequential(OrderedDict([("mu_mlp0", nn.Linear(in_features, latent_space_size, bias=True))]))
        if not easy_init:
            with torch.no_grad():
                self.mus[-1].weight.normal_(std=mu_weight_std)
                nn.init.constant_(self.mus[-1].bias, mu_bias)
        if not deterministic:
            self.logvars = nn.Sequential(OrderedDict([("logvar_mlp0", nn.Linear(in_features, latent_space_size, bias=True))]))
            if not easy_init:
                with torch.no_grad():
                    self.logvars[-1].weight.normal_(std=logvar_weight_std)
                    nn.init.constant_(self.logvars[-1].bias, logvar_bias)

    def forward(self, input):
        features = self.features(input) if self.n_layers > 0 else input
        if self.deterministic:
            return self.mus(features)
        return self.mus(features), self.logvars(features)
