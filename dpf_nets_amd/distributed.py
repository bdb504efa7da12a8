"""Data parallelism for the hot path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed code.  The path shards by
cloud -- every op is independent per cloud -- so inference/evaluation needs no
data-path collective: each rank takes a contiguous slice of the batch.  Training
adds exactly ONE all-reduce of the flattened gradient per step, between
`loss.backward()` (lib/networks/training.py:55) and `optimizer.step()` (:56).
BatchNorm statistics stay per replica (no SyncBN), so parity is per shard."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items, rank=None, world_size=None):
    """Contiguous [lo, hi) slice of n_items for this rank; sizes differ by at most one."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(*tensors):
    """Slice every tensor along dim 0 to this rank's clouds."""
    lo, hi = shard_bounds(tensors[0].shape[0])
    out = tuple(t[lo:hi].contiguous() for t in tensors)
    return out if len(out) > 1 else out[0]


def gather_clouds(t):
    """All-gather per-cloud results (e.g. the (B_local,) CD vector) back into batch order;
    shards may differ in size by one cloud."""
    rank, w = world()
    if w == 1:
        return t
    n_local = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n_local) for _ in range(w)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = t.new_zeros((m,) + tuple(t.shape[1:]))
    pad[:t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def allreduce_gradients(parameters, average=True):
    """One collective per step: flatten every .grad into a single fp32 buffer, all-reduce it
    (sum), divide by the world size, scatter back.  13 M parameters = 52 MB for the all_scaled
    model -- one bucket, so RCCL sees one large message per step instead of ~900 small ones."""
    rank, w = world()
    # every rank walks ALL trainable parameters in the caller's order, so the flat buffer has the same length
    # and layout everywhere even if a sub-module produced no gradient on some shard (a missing .grad counts as
    # zeros and receives the other ranks' average)
    params = [p for p in parameters if p.requires_grad]
    if w == 1 or not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(w)
    off = 0
    for p in params:
        n = p.numel()
        piece = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = piece.clone()
        else:
            p.grad.copy_(piece)
        off += n
    return flat.numel()


def allreduce_flat_gradients(store, average=True):
    """The same single collective for a decoder whose parameters live in a FlatStore
    (LocalCondRNVPDecoder.flatten_parameters): the gradient buffer IS flat, so there is nothing to
    gather or scatter -- one all-reduce of `store.flat_g` in place."""
    rank, w = world()
    if w == 1:
        return 0
    store.attach_grads()
    store.grad_written = True            # the other ranks' average is a gradient even if this shard produced none
    dist.all_reduce(store.flat_g, op=dist.ReduceOp.SUM)
    if average:
        store.flat_g.div_(w)
    return store.flat_g.numel()


def broadcast_buffers(module, src=0):
    """BatchNorm running statistics diverge across replicas (no SyncBN); broadcast rank `src`'s
    before a checkpoint so that every rank saves the same state dict."""
    rank, w = world()
    if w == 1:
        return
    for b in module.buffers():
        if b.dtype.is_floating_point:
            dist.broadcast(b, src)
