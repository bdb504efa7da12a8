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
    """(N1, N2) matrix of Chamfer distances, cds[i, j] = CD(clouds1[i], clouds2[j])
    (lib/networks/utils.py:90-117).  Row i is ONE strided Chamfer launch (matrix-core filtered for big rows, same
    bits) -- cloud i is broadcast
    against the batch by a zero stride instead of being expanded and copied N2 times -- plus one
    reduction launch; results are those of the reference's expand-and-call loop."""
    from .._lib import lib, check, current_stream
    if not (clouds1.is_cuda and clouds2.is_cuda):
        raise RuntimeError("pairwise_CD needs CUDA tensors")
    clouds1, clouds2 = clouds1.contiguous(), clouds2.contiguous()
    N1, n = clouds1.shape[0], clouds1.shape[1]
    N2, m = clouds2.shape[0], clouds2.shape[1]
    dev = clouds1.device
    cds = torch.empty((N1, N2), dtype=torch.float32, device=dev)
    bs = max(1, min(bs, N2))
    d1 = torch.empty((bs, n), dtype=torch.float32, device=dev)
    d2 = torch.empty((bs, m), dtype=torch.float32, device=dev)
    i1 = torch.empty((bs, n), dtype=torch.int32, device=dev)
    i2 = torch.empty((bs, m), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        for i in range(N1):
            for j_l in range(0, N2, bs):
                nb = min(N2, j_l + bs) - j_l
                check(lib().dpf_nndistance_strided_auto(nb, n, clouds1[i].data_ptr(), 0, m, clouds2[j_l].data_ptr(), m * 3,
                                                   d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), st),
                      "nndistance_strided")
                check(lib().dpf_chamfer_reduce(nb, n, m, d1.data_ptr(), d2.data_ptr(),
                                               cds[i, j_l:j_l + nb].data_ptr(), st), "chamfer_reduce")
    return cds


def emd_approx(sample, ref):
    B, N, N_ref = sample.size(0), sample.size(1), ref.size(1)
    assert N == N_ref, "Not sure what would EMD do in this case"
    return match_cost(sample, ref) / float(N)
