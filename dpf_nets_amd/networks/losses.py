"""PointFlowNLL (lib/networks/losses.py:7-15).

The reference evaluates `sum(logvars)` as 63 elementwise adds over the decoder's
list.  When the list is (or ends with the views of) a FlowList produced by the
fused HIP stack, the layer-sum was already accumulated in registers by the
kernel and is used directly."""
import math

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


class PointFlowNLL(nn.Module):
    def forward(self, samples, mus, logvars):
        s0 = samples[0]
        tot = total_logvar(logvars) + (s0 - mus[0]) ** 2 / torch.exp(logvars[0])
        return 0.5 * (tot.sum() / s0.shape[0] + math.log(2.0 * math.pi) * s0.shape[1] * s0.shape[2])
