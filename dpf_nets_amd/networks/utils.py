"""Chamfer call sites of the reference's lib/networks/utils.py (metrics half):
distChamferCUDA (:34-35), f_score (:38-42), pairwise_CD (:90-117), and
emd_approx (lib/metrics/evaluation_metrics.py:26-31)."""
import torch

from ..metrics.StructuralLosses.nn_distance import nn_distance
from ..metrics.StructuralLosses.match_cost import match_cost


def distChamferCUDA(x, y):
    return nn_distance(x, y)


def chamfer_distance(pred, true):
    """cd = (dl.mean(1) + dr.mean(1)).mean() as evaluating.py:110-113 reduces it."""
    dl, dr = distChamferCUDA(pred, true)
    return (dl.mean(1) + dr.mean(1)).mean()


def chamfer_per_cloud(dl, dr):
    """(B,) dl.mean(1) + dr.mean(1) in one deterministic HIP launch (evaluating.py:112)."""
    from .._lib import lib, check, current_stream
    if not (dl.is_cuda and dr.is_cuda and dl.is_contiguous() and dr.is_contiguous()):
        raise RuntimeError("dl, dr must be contiguous CUDA tensors")
    cd = torch.empty((dl.shape[0],), dtype=torch.float32, device=dl.device)
    with torch.cuda.device(dl.device):
        check(lib().dpf_chamfer_reduce(dl.shape[0], dl.shape[1], dr.shape[1], dl.data_ptr(), dr.data_ptr(),
                                       cd.data_ptr(), current_stream()), "chamfer_reduce")
    return cd


def f_score(predicted_clouds, true_clouds, threshold=0.001):
    ld, rd = distChamferCUDA(predicted_clouds, true_clouds)
    precision = 100.0 * (rd < threshold).float().mean(1)
    recall = 100.0 * (ld < threshold).float().mean(1)
    return 2.0 * precision * recall / (precision + recall + 1e-7)


def pairwise_CD(clouds1, clouds2, bs=2048):
    """(N1, N2) matrix of Chamfer distances; row i = cloud1[i] against every cloud2."""
    N1, N2 = clouds1.shape[0], clouds2.shape[0]
    cds = torch.zeros((N1, N2), dtype=torch.float32, device=clouds1.device)
    for i in range(N1):
        for j_l in range(0, N2, bs):
            j_u = min(N2, j_l + bs)
            c1 = clouds1[i].unsqueeze(0).expand(j_u - j_l, -1, -1).contiguous()
            dl, dr = distChamferCUDA(c1, clouds2[j_l:j_u].contiguous())
            cds[i, j_l:j_u] = dl.mean(dim=1) + dr.mean(dim=1)
    return cds


def emd_approx(sample, ref):
    B, N, N_ref = sample.size(0), sample.size(1), ref.size(1)
    assert N == N_ref, "Not sure what would EMD do in this case"
    return match_cost(sample, ref) / float(N)
