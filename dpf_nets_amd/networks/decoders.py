"""LocalCondRNVPDecoder with the reference's constructor and forward() contract
(lib/networks/decoders.py:41-72): forward(p, g, mode) -> (ps, mus, logvars), three
list-likes of 3*n_flows (B,3,N) tensors in DIRECT order for both modes.

Eval mode runs ALL layers in one fused HIP launch (csrc/flow.hip); the lists are
FlowLists over three (L,B,3,N) buffers the kernel fills.  Training mode on CUDA
tensors runs the HIP training kernels for the whole stack as one autograd node
(networks/train_engine.py); `forward_torch` chains the layers' tensor-op path
(decoders.py:58-70) for CPU tensors / DPF_TRAIN_IMPL=torch."""
import torch.nn as nn

from .flows import CondRealNVPFlow3DTriple, _needs_autograd, use_hip_training, train_stack, stack_spec
from .flowlist import FlowList
from .engine import FlowStack


class LocalCondRNVPDecoder(nn.Module):
    def __init__(self, n_flows, f_n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows, self.f_n_features, self.g_n_features = n_flows, f_n_features, g_n_features
        self.weight_std = weight_std
        self.flows = nn.ModuleList([
            CondRealNVPFlow3DTriple(f_n_features, g_n_features, weight_std=weight_std, pattern=i % 2)
            for i in range(n_flows)])
        object.__setattr__(self, "_stack", None)
        self.precision = None          # None -> engine.DEFAULT_PRECISION ("f16x3")
        self.materialize_lists = True  # False: skip the 3 x L per-layer tensors (lists then hold the final layer only)
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate_packed())

    def coupling_layers(self):
        """All 3*n_flows CondRealNVPFlow3D modules in DIRECT order."""
        return [lyr for tri in self.flows for lyr in tri.layers()]

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

    def stack(self):
        if self._stack is None:
            object.__setattr__(self, "_stack", FlowStack(self.coupling_layers()))
        return self._stack

    def flatten_parameters(self):
        """Opt-in (extension; DPF_TRAIN_FLAT=1 does it on the first training step): move every parameter and
        BatchNorm buffer of the coupling layers into ONE flat buffer laid out as the HIP training path consumes it,
        with `.grad` as views of a twin gradient buffer (train_engine.FlatStore).  Names, shapes, state dicts and
        optimizers are unaffected; the training step loses the per-parameter gather, scatter and AccumulateGrad
        work (2016 tensors at n_flows=21).  Call it after the module is on its GPU.  Returns the store
        (`.flat_p`, `.flat_g`, `.zero_grad()`); data-parallel runs reduce `.flat_g` with
        dpf_nets_amd.distributed.allreduce_flat_gradients instead of DistributedDataParallel hooks."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("flatten_parameters: move the decoder to its GPU first")
        return stack_spec(self, self.coupling_layers()).flatten(dev)

    def flat_store(self):
        """The FlatStore of the full stack, or None."""
        spec = self.__dict__.get("_train_specs", {}).get(len(self.coupling_layers()))
        return None if spec is None else spec.flat

    def forward_torch(self, p, g, mode="direct"):
        ps, mus, lvs = [], [], []
        for i in range(self.n_flows):                               # decoders.py:58-70
            if mode == "direct":
                buf = self.flows[i].forward_torch(p if i == 0 else ps[-1], g, mode=mode)
                ps, mus, lvs = ps + buf[0], mus + buf[1], lvs + buf[2]
            else:
                buf = self.flows[-(i + 1)].forward_torch(p if i == 0 else ps[0], g, mode=mode)
                ps, mus, lvs = buf[0] + ps, buf[1] + mus, buf[2] + lvs
        return ps, mus, lvs

    def sample_and_decode(self, mu0, logvar0, g, noise=None):
        """`reparameterize` + `forward(z, g, mode='direct')` of lib/networks/models.py:211-213 (and :118-119, :250-253) as
        ONE call (extension): returns (z, ps, mus, logvars) with z = noise * exp(0.5 * logvar0) + mu0 =
        what the models store as p_prior_samples[0].  mu0 / logvar0 are the models' (B,3,S) stride-0 expansions, read in
        place; `noise` defaults to torch.randn_like(logvar0) -- drawn by torch, so the generator stream is the caller's.
        In eval mode on CUDA the sample is formed in the fused kernel's prologue (no exp / mul / add launches, no
        materialised z round trip); otherwise this is exactly the two reference calls."""
        import torch
        if noise is None:
            noise = torch.randn_like(logvar0)                                      # models.py:78
        fused = (not self.training and noise.is_cuda and not _needs_autograd(noise, mu0, logvar0, g)
                 and noise.dtype == torch.float32 and mu0.dtype == torch.float32 and logvar0.dtype == torch.float32)
        if not fused:
            z = noise.mul(torch.exp(0.5 * logvar0)).add_(mu0)                      # models.py:77-79
            return (z,) + tuple(self.forward(z, g, mode="direct"))
        stack = self.stack()
        p_out, sum_lv, ps, mus, lvs = stack.run(noise, g, "direct", self.precision, want_lists=self.materialize_lists,
                                                base=(mu0, logvar0))
        z = stack.last_base_sample
        if ps is None:
            return z, FlowList(p_out.unsqueeze(0)), None, FlowList(sum_lv.unsqueeze(0), sum_lv)
        return z, FlowList(ps), FlowList(mus), FlowList(lvs, sum_lv)

    def forward(self, p, g, mode="direct", n_layers=None):
        """n_layers (extension, default all): run only the first n_layers DIRECT-order layers
        (the BASELINE metric's 14-layer stack = first 14 layers of n_flows=5)."""
        if mode not in ("direct", "inverse"):
            raise ValueError(mode)
        if use_hip_training(self, p):
            layers = self.coupling_layers()
            if n_layers is not None:
                layers = layers[:int(n_layers)]
            ps, mus, lvs = train_stack(self, layers, p, g, mode, allow_flat=n_layers is None)
            return ps, mus, lvs
        if self.training or _needs_autograd(p, g):
            if n_layers is not None:
                raise ValueError("n_layers is only supported on the fused eval path")
            return self.forward_torch(p, g, mode)
        p_out, sum_lv, ps, mus, lvs = self.stack().run(p, g, mode, self.precision,
                                                       want_lists=self.materialize_lists, n_layers=n_layers)
        if ps is None:
            return FlowList(p_out.unsqueeze(0)), None, FlowList(sum_lv.unsqueeze(0), sum_lv)
        return FlowList(ps), FlowList(mus), FlowList(lvs, sum_lv)
