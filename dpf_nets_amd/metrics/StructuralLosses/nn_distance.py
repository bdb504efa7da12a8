"""nn_distance(seta (B,n,3), setb (B,m,3)) -> (dist1 (B,n), dist2 (B,m)).
Mirror of lib/metrics/pytorch_structural_losses/nn_distance.py:7-41."""
import torch

from .StructuralLossesBackend import NNDistance, NNDistanceGrad


class NNDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seta, setb):
        ctx.save_for_backward(seta, setb)
        dist1, idx1, dist2, idx2 = NNDistance(seta, setb)
        ctx.idx1, ctx.idx2 = idx1, idx2          # indices ride on ctx, as in the reference (:22-23)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        seta, setb = ctx.saved_tensors
        grada, gradb = NNDistanceGrad(seta, setb, ctx.idx1, ctx.idx2,
                                      grad_dist1.contiguous(), grad_dist2.contiguous())
        return grada, gradb


nn_distance = NNDistanceFunction.apply
