"""PointFlowNLL (lib/networks/losses.py:7-15).

The reference evaluates `sum(logvars)` as 63 elementwise adds over the decoder's
list.  When the list is (or ends with the views of) a FlowList produced by the
fused HIP stack, the layer-sum was already accumulated in registers by the
kernel and is used directly."""
import math

import numpy as np

import torch
import torch.nn as nn

from .flowlist import FlowList


def total_logvar(logvars):
    """sum(logvars) for a plain list, a FlowList, or `[prior...] + FlowList` lists."""
    if isinstance(logvars, FlowList):
        return logvars.total()
    tag = getattr(logvars[-1], "_dpf_total", None) if len(logvars) else None
    if tag is not None:
        token, k, total = tag
        if len(logvars) >= k and all(
                getattr(logvars[len(logvars) - k + i], "_dpf_pos", None) == (token, i) for i in range(k)):
            head = logvars[:len(logvars) - k]
            return (sum(head) + total) if len(head) else total
    return sum(logvars)


def _flow_total(logvars):
    """(total of logvars[1:], True) when logvars is `[base] + <the fused stack's list>` and the stack left its layer-sum."""
    if len(logvars) < 2:
        return None, False
    tag = getattr(logvars[-1], "_dpf_total", None)
    if tag is None:
        return None, False
    token, k, total = tag
    if len(logvars) == k + 1 and all(getattr(logvars[1 + i], "_dpf_pos", None) == (token, i) for i in range(k)):
        return total, True
    return None, False


class _PointFlowNLLNode(torch.autograd.Function):
    """losses.py:11-15 as ONE autograd node over HIP kernels (training step): forward =
    dpf_pointflow_nll (one pass over s0 and the stack's layer-sum of log-variances, base distribution through its strides),
    backward = dpf_pointflow_nll_backward (one launch: d s0, the constant d sum_lv, and d mu0 / d lv0 only if the base
    distribution is learned)."""

    @staticmethod
    def forward(ctx, s0, mu0, lv0, total):
        from .._lib import lib, check, current_stream
        B, C, N = s0.shape
        s0c = s0.contiguous()
        out = torch.empty((), dtype=torch.float32, device=s0.device)
        ws = torch.empty(lib().dpf_pointflow_nll_workspace_floats(), dtype=torch.float32, device=s0.device)
        with torch.cuda.device(s0.device):
            check(lib().dpf_pointflow_nll(B, C, N, s0c.data_ptr(), mu0.data_ptr(), *mu0.stride(), lv0.data_ptr(), *lv0.stride(),
                                          total.data_ptr(), ws.data_ptr(), out.data_ptr(), current_stream()), "pointflow_nll")
        ctx.save_for_backward(s0c, mu0, lv0)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from .._lib import lib, check, current_stream
        s0, mu0, lv0 = ctx.saved_tensors
        B, C, N = s0.shape
        need = ctx.needs_input_grad
        d_s0 = torch.empty_like(s0) if need[0] else None
        d_mu = torch.empty_like(s0) if need[1] else None
        d_lv = torch.empty_like(s0) if need[2] else None
        d_tot = torch.empty_like(s0) if need[3] else None
        g = grad_out.contiguous().to(torch.float32)
        with torch.cuda.device(s0.device):
            check(lib().dpf_pointflow_nll_backward(B, C, N, s0.data_ptr(), mu0.data_ptr(), *mu0.stride(), lv0.data_ptr(), *lv0.stride(),
                                                   g.data_ptr(), d_s0.data_ptr() if need[0] else None,
                                                   d_tot.data_ptr() if need[3] else None, d_mu.data_ptr() if need[1] else None,
                                                   d_lv.data_ptr() if need[2] else None, current_stream()), "pointflow_nll_backward")
        return d_s0, d_mu, d_lv, d_tot


class PointFlowNLL(nn.Module):
    def forward(self, samples, mus, logvars):
        s0, mu0, lv0 = samples[0], mus[0], logvars[0]
        # training step on CUDA tensors: the stack's layer-sum of log-variances is at hand -> one fused node
        if s0.is_cuda and s0.dim() == 3 and s0.dtype == torch.float32 and mu0.shape == s0.shape and lv0.shape == s0.shape and \
                mu0.dtype == lv0.dtype == torch.float32 and torch.is_grad_enabled() and \
                (s0.requires_grad or mu0.requires_grad or lv0.requires_grad):
            total, ok = _flow_total(logvars)
            if ok and total.dtype == torch.float32 and total.shape == s0.shape:
                return _PointFlowNLLNode.apply(s0, mu0, lv0, total.contiguous())
        # evaluation (CUDA tensors, nothing to differentiate): one pass over s0 and the kernel's sum of log-variances, the
        # base distribution's stride-0 expansions read through their strides (csrc/nll.hip)
        if s0.is_cuda and s0.dim() == 3 and s0.dtype == torch.float32 and mu0.shape == s0.shape and lv0.shape == s0.shape and \
                mu0.dtype == lv0.dtype == torch.float32 and \
                not (torch.is_grad_enabled() and (s0.requires_grad or mu0.requires_grad or lv0.requires_grad)):
            total, ok = _flow_total(logvars)
            if ok and total.is_contiguous() and not (torch.is_grad_enabled() and total.requires_grad):
                from .._lib import lib, check, current_stream
                B, C, N = s0.shape
                s0 = s0.contiguous()
                out = torch.empty((), dtype=torch.float32, device=s0.device)
                ws = torch.empty(lib().dpf_pointflow_nll_workspace_floats(), dtype=torch.float32, device=s0.device)
                with torch.cuda.device(s0.device):
                    check(lib().dpf_pointflow_nll(B, C, N, s0.data_ptr(), mu0.data_ptr(), *mu0.stride(), lv0.data_ptr(), *lv0.stride(),
                                                  total.data_ptr(), ws.data_ptr(), out.data_ptr(), current_stream()), "pointflow_nll")
                return out
        tot = total_logvar(logvars) + (s0 - mu0) ** 2 / torch.exp(lv0)
        return 0.5 * (tot.sum() / s0.shape[0] + math.log(2.0 * math.pi) * s0.shape[1] * s0.shape[2])


class GaussianFlowNLL(nn.Module):
    """lib/networks/losses.py:18-26 -- O(B*G) tensor ops on the latent prior flow's lists."""

    def forward(self, samples, mus, logvars):
        # total_logvar: sum(logvars), taken from the stack's own layer-sum when the list is `[base] + <the prior flow's list>`
        return 0.5 * torch.add(torch.sum(total_logvar(logvars) + ((samples[0] - mus[0]) ** 2 / torch.exp(logvars[0]))) / samples[0].shape[0],
                               np.log(2.0 * np.pi) * samples[0].shape[1])


class GaussianEntropy(nn.Module):
    """lib/networks/losses.py:29-34"""

    def forward(self, logvars):
        return 0.5 * torch.add(logvars.shape[1] * (1.0 + np.log(2.0 * np.pi)), logvars.sum(1).mean())


class Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(nn.Module):
    """lib/networks/losses.py:37-51: (weighted total, pnll, gnll, gent)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.pnll_weight, self.gnll_weight, self.gent_weight = kwargs.get("pnll_weight"), kwargs.get("gnll_weight"), kwargs.get("gent_weight")
        self.PNLL, self.GNLL, self.GENT = PointFlowNLL(), GaussianFlowNLL(), GaussianEntropy()

    def forward(self, g_clouds, l_clouds, outputs):
        pnll = self.PNLL(outputs["p_prior_samples"], outputs["p_prior_mus"], outputs["p_prior_logvars"])
        gnll = self.GNLL(outputs["g_prior_samples"], outputs["g_prior_mus"], outputs["g_prior_logvars"])
        gent = self.GENT(outputs["g_posterior_logvars"])
        return self.pnll_weight * pnll + self.gnll_weight * gnll - self.gent_weight * gent, pnll, gnll, gent
