"""SharedDot / Swish with the reference's parameter names, shapes and init
(lib/networks/layers.py:5-45).  These modules carry the weights (state-dict
compatibility) and serve the training-mode path; in eval mode the coupling
layers read the weights out of them and run the fused HIP stack instead."""
import torch
import torch.nn as nn


class Swish(nn.Module):
    def forward(self, x):                       # layers.py:9-10
        return x * torch.sigmoid(x)


class SharedDot(nn.Module):
    """The same (out x in) linear map applied to every point: weight is
    (n_channels, out, in), input (B, in, N) -> (B, out, N)."""

    def __init__(self, in_features, out_features, n_channels, bias=False, init_weight=None, init_bias=None):
        super().__init__()
        self.in_features, self.out_features, self.n_channels = in_features, out_features, n_channels
        self.init_weight, self.init_bias = init_weight, init_bias
        self.weight = nn.Parameter(torch.empty(n_channels, out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(n_channels, out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):                 # layers.py:29-38
        if self.init_weight:
            nn.init.uniform_(self.weight.data, a=-self.init_weight, b=self.init_weight)
        else:
            # on the 3-D weight fan_in = out*in (n_channels is dim 0): bound sqrt(6/(out*in))
            nn.init.kaiming_uniform_(self.weight.data, a=0.0)
        if self.bias is not None:
            nn.init.constant_(self.bias.data, self.init_bias if self.init_bias else 0.0)

    def forward(self, x):                       # layers.py:40-45
        out = torch.matmul(self.weight, x.unsqueeze(1))
        if self.bias is not None:
            out = out + self.bias.unsqueeze(0).unsqueeze(3)
        return out.squeeze(1)
