"""match_cost(seta (B,n,3), setb (B,m,3)) -> cost (B,): approximate EMD.
Mirror of lib/metrics/pytorch_structural_losses/match_cost.py:6-44 (the matching
is a constant in backward, :38-42)."""
import torch

from .StructuralLossesBackend import ApproxMatch, ApproxMatchCost, MatchCost, MatchCostGrad  # noqa: F401


class MatchCostFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        match, _temp, cost = ApproxMatchCost(seta, setb)        # = ApproxMatch, then MatchCost (match_cost.py:20-22)
        ctx.match = match
        return cost

    @staticmethod
    def backward(ctx, grad_output):
        seta, setb = ctx.saved_tensors
        grada, gradb = MatchCostGrad(seta, setb, ctx.match)
        go = grad_output.unsqueeze(1).unsqueeze(2)
        return grada * go, gradb * go


match_cost = MatchCostFunction.apply
