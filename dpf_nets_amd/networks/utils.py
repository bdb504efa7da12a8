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


class ChamferEvaluator:
    """The evaluation loop's Chamfer call (evaluating.py:110-113: `nn_distance` both ways, then `dl.mean(1) + dr.mean(1)`) as
    ONE launch per batch: distances, indices and the per-cloud CD from the search kernel's own workgroups (NNDistanceCD).  The
    evaluator OWNS the scratch that call needs -- one CDWorkspace per (batch shape, device, stream), created outside any stream
    capture, as the interface's ownership rule asks (StructuralLossesBackend.CDWorkspace) -- so a caller that keeps one
    evaluator per evaluation loop gets the single-launch path with nothing cached behind its back.  r05: the
    package's own evaluation helpers (metrics.evaluation_metrics.EMD_CD, chamfer_cd_per_cloud) and bench.py's step go through
    this class; before, only the benchmark created a workspace."""

    MAX_WORKSPACES = 8          # (a ragged last batch and every new shape used to add a buffer for good)

    def __init__(self):
        import collections
        import threading
        self._ws = collections.OrderedDict()            # least recently used first
        self._lock = threading.Lock()

    def __call__(self, pred, true):
        """(B, n, 3), (B, m, 3) contiguous CUDA tensors -> dist1, idx1, dist2, idx2, cd (B,)"""
        from ..metrics.StructuralLosses import StructuralLossesBackend as BK
        # keyed by the stream OBJECT (held alive by the key: its handle cannot be recycled for another stream while the
        # workspace lives) and the calling thread (two threads on one stream would share tickets)
        import threading
        stream = torch.cuda.current_stream(pred.device)
        key = (pred.shape[0], pred.shape[1], true.shape[1], pred.device, stream, threading.get_ident())
        with self._lock:
            ws = self._ws.get(key)
            if ws is not None:
                self._ws.move_to_end(key)
            elif not torch.cuda.is_current_stream_capturing():
                ws = self._ws[key] = BK.CDWorkspace(key[0], key[1], key[2], pred.device)
                while len(self._ws) > self.MAX_WORKSPACES:
                    self._ws.popitem(last=False)
        return BK.NNDistanceCD(pred, true, ws)        # (no workspace while capturing a first call: fresh scratch, tickets cleared in-call)

    def cd(self, pred, true):
        return self(pred, true)[4]


_default_evaluator = None


def chamfer_cd_per_cloud(pred, true):
    """(B,) Chamfer distance of each pair of clouds, cd[b] = dist1[b].mean() + dist2[b].mean(), in one launch (no autograd:
    the evaluation path).  Uses a process-wide ChamferEvaluator; loops that run on several streams should keep their own."""
    global _default_evaluator
    if _default_evaluator is None:
        _default_evaluator = ChamferEvaluator()
    return _default_evaluator.cd(pred, true)


def f_score(predicted_clouds, true_clouds, threshold=0.001):
    """lib/networks/utils.py:38-42.  Without autograd the thresholds, means and the F1 formula are one launch over the two
    distance rows (dpf_fscore_reduce: integer counts, the float arithmetic in the reference's order)."""
    ld, rd = distChamferCUDA(predicted_clouds, true_clouds)
    if ld.is_cuda and not (torch.is_grad_enabled() and (ld.requires_grad or rd.requires_grad)):
        from .._lib import lib, check, current_stream
        out = torch.empty((ld.shape[0],), dtype=torch.float32, device=ld.device)
        ldc, rdc = ld.contiguous(), rd.contiguous()                       # locals: they must outlive the launch's enqueue
        with torch.cuda.device(ld.device):
            check(lib().dpf_fscore_reduce(ld.shape[0], ld.shape[1], rd.shape[1], ldc.data_ptr(), rdc.data_ptr(),
                                          float(threshold), out.data_ptr(), current_stream()), "fscore_reduce")
        return out
    precision = 100.0 * (rd < threshold).float().mean(1)
    recall = 100.0 * (ld < threshold).float().mean(1)
    return 2.0 * precision * recall / (precision + recall + 1e-7)


def pairwise_CD(clouds1, clouds2, bs=2048, shard_rows=False):
    """(N1, N2) matrix of Chamfer distances, cds[i, j] = CD(clouds1[i], clouds2[j]) (lib/networks/utils.py:90-117; three
    calls per generative evaluation, evaluating.py:245-247).

    ONE launch for the whole matrix (dpf_pairwise_cd: grid over (query tile, j, 2 i + direction), both clouds read in
    place, the per-point distances summed in the kernel) plus a finish over the workgroups' fixed-order partial sums --
    instead of the reference's loop of N1 `expand + contiguous + nn_distance + mean` calls.  `bs` (the reference's
    batching of the second set) only bounds the workspace here: rows are processed in chunks of at most bs pairs' worth.
    shard_rows: under torch.distributed every rank computes its contiguous block of rows (distributed.shard_bounds) and
    the (N1, N2) matrix is all-gathered -- no other communication.
    Clouds of fewer than 32 points fall back to one strided launch + one reduction per row."""
    from .._lib import lib, check, current_stream
    if not (clouds1.is_cuda and clouds2.is_cuda):
        raise RuntimeError("pairwise_CD needs CUDA tensors")
    clouds1, clouds2 = clouds1.contiguous(), clouds2.contiguous()
    if clouds1.dtype != torch.float32 or clouds2.dtype != torch.float32:
        raise RuntimeError("pairwise_CD needs float32 clouds")
    N1, n = clouds1.shape[0], clouds1.shape[1]
    N2, m = clouds2.shape[0], clouds2.shape[1]
    dev = clouds1.device
    lo, hi = 0, N1
    if shard_rows:
        from .. import distributed as D
        lo, hi = D.shard_bounds(N1)
    cds = torch.empty((hi - lo, N2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = current_stream()
        if n >= 32 and m >= 32 and N2 <= 65535:
            rows = max(1, min(hi - lo, 32767, max(1, (bs * 64) // max(N2, 1))))
            nbytes = lib().dpf_pairwise_cd_workspace_bytes(rows, N2, n, m)
            ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
            for r0 in range(lo, hi, rows):
                nr = min(rows, hi - r0)
                check(lib().dpf_pairwise_cd(nr, N2, n, m, clouds1[r0].data_ptr(), clouds2.data_ptr(), cds[r0 - lo].data_ptr(),
                                            ws.data_ptr(), nbytes, st), "pairwise_cd")
        else:
            bs = max(1, min(bs, N2))
            d1 = torch.empty((bs, n), dtype=torch.float32, device=dev)
            d2 = torch.empty((bs, m), dtype=torch.float32, device=dev)
            i1 = torch.empty((bs, n), dtype=torch.int32, device=dev)
            i2 = torch.empty((bs, m), dtype=torch.int32, device=dev)
            for i in range(lo, hi):
                for j_l in range(0, N2, bs):
                    nb = min(N2, j_l + bs) - j_l
                    check(lib().dpf_nndistance_strided_auto(nb, n, clouds1[i].data_ptr(), 0, m, clouds2[j_l].data_ptr(), m * 3,
                                                            d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), st),
                          "nndistance_strided")
                    check(lib().dpf_chamfer_reduce(nb, n, m, d1.data_ptr(), d2.data_ptr(),
                                                   cds[i - lo, j_l:j_l + nb].data_ptr(), st), "chamfer_reduce")
    if shard_rows:
        from .. import distributed as D
        cds = D.gather_clouds(cds)
    return cds


def emd_approx(sample, ref):
    B, N, N_ref = sample.size(0), sample.size(1), ref.size(1)
    assert N == N_ref, "Not sure what would EMD do in this case"
    return match_cost(sample, ref) / float(N)


# ----------------------------------------------------------------------------------------------------------------
# The consumers of pairwise_CD in the reference's generative evaluation (evaluating.py:245-262): coverage, minimum matching
# distance and the 1-NN two-sample accuracy over the (N1, N2) Chamfer matrices, the voxel-occupancy Jensen-Shannon divergence
# over the clouds themselves, and the running-average helper of both loops.  Same names, arguments and values as
# lib/networks/utils.py:8-22, 45-87, 120-146; tensor / numpy plumbing, no kernels.
# ----------------------------------------------------------------------------------------------------------------
class AverageMeter:
    """lib/networks/utils.py:8-22: last value, running sum / count / mean."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def COV(dists, axis=1):
    """Coverage (utils.py:120-121): the share of the clouds along `axis` that are the nearest neighbour of some cloud of the
    other set."""
    nearest = torch.argmin(dists, dim=axis)
    return float(torch.unique(nearest).numel()) / float(dists.shape[axis])


def MMD(dists, axis=1):
    """Minimum matching distance (utils.py:124-125): every cloud along `axis` to its nearest cloud of the other set, averaged."""
    return float(torch.amin(dists, dim=(axis + 1) % 2).float().mean())


def KNN(Mxx, Mxy, Myy, k, sqrt=False):
    """Leave-one-out k-NN two-sample accuracy (utils.py:128-146): the two sets are labelled -1 / +1, every cloud is classified
    by the sign of the label sum of its k nearest other clouds (ties go to +1); returns the share classified correctly."""
    n0, n1 = Mxx.shape[0], Myy.shape[0]
    label = torch.cat((-torch.ones(n0), torch.ones(n1))).to(Mxx)
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.t(), Myy), 1)), 0)
    if sqrt:
        M = M.abs().sqrt()
    M = M + torch.diag(torch.full((n0 + n1,), float("inf")).to(Mxx))       # a cloud is not its own neighbour
    idx = torch.topk(M, k, dim=0, largest=False).indices                   # (k, n0 + n1): neighbours of every column
    votes = label[idx].sum(0)
    pred = torch.where(votes >= 0, torch.ones_like(votes), -torch.ones_like(votes))
    return float((pred == label).float().mean())


def get_voxel_occ_dist(all_clouds, clouds_flag='gen', res=28, bound=0.5, bs=128, warning=True):
    """Occupancy distribution of the points of all clouds over the res^3 voxels of [-0.5, 0.5)^3 (utils.py:45-80).
    all_clouds: (K, n, 3) numpy array.  A point outside the cube (or NaN) is not counted.  The bin edges are the reference's
    doubles -0.5 + i / res and the intervals half-open on the right, so every point lands in the voxel the reference
    picks; the counting is one bincount instead of batched comparison tables (`bs` is accepted and ignored)."""
    import numpy as np
    all_clouds = np.asarray(all_clouds)
    if np.any(np.fabs(all_clouds) > bound) and warning:
        print('{} clouds out of cube bounds: [-{}; {}]'.format(clouds_flag, bound, bound))
    n_nans = np.isnan(all_clouds).sum()
    if n_nans > 0:
        print('{} NaN values in point cloud tensors.'.format(n_nans))
    edges = -0.5 + np.arange(res + 1) * (1. / res)
    pts = all_clouds.reshape(-1, 3)
    cell = np.searchsorted(edges, pts, side='right') - 1                    # edges[c] <= x < edges[c + 1]; NaN sorts past the end
    inside = np.all((cell >= 0) & (cell < res), axis=1)
    cell = cell[inside]
    flat = (cell[:, 0] * res + cell[:, 1]) * res + cell[:, 2]
    counts = np.bincount(flat, minlength=res ** 3).astype(np.uint64).reshape(res, res, res)
    return np.float64(counts) / counts.sum()


def JSD(clouds1, clouds2, clouds1_flag='gen', clouds2_flag='ref', warning=True):
    """Jensen-Shannon divergence (base 2) between the voxel occupancy distributions of two sets of clouds (utils.py:83-87)."""
    from scipy.stats import entropy
    d1 = get_voxel_occ_dist(clouds1, clouds_flag=clouds1_flag, warning=warning).flatten()
    d2 = get_voxel_occ_dist(clouds2, clouds_flag=clouds2_flag, warning=warning).flatten()
    return entropy((d1 + d2) / 2.0, base=2) - 0.5 * (entropy(d1, base=2) + entropy(d2, base=2))


def save_model(state, model_name):
    """utils.py:25-27 (training.py's checkpoint writer): pickle protocol 4, one line on stdout."""
    torch.save(state, model_name, pickle_protocol=4)
    print('Model saved to ' + model_name)


def cnt_params(params):
    """utils.py:30-31: number of trainable scalars."""
    return sum(p.numel() for p in params if p.requires_grad)
