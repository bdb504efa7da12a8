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


class GradArena:
    """ONE flat fp32 gradient buffer for a whole model -- what makes "a single all-reduce per step" literal.

    Layout: [flat_g of every FlatStore / PriorFlatStore among the parameters (the point decoder, the latent prior flow) |
    every other trainable parameter (encoder, posterior / prior heads, g0 prior ...) in the caller's order].  The stores'
    gradient buffers are re-based onto their slice (FlatStore.rebase_grads); every other parameter's `.grad` becomes a
    view of its slice, so autograd accumulates straight into the message.  `allreduce()` is then exactly one
    `dist.all_reduce` of `buf` (sum, / world): 52 MB for the all_scaled model (12 972 413 parameters + 4 KB of layout
    padding), nothing gathered or scattered.  Use `arena.zero_grad()` (one fill, views stay attached) in place of
    `optimizer.zero_grad()`; a `.grad` that was dropped or replaced behind the arena's back (set_to_none) is re-attached
    by `sync()`, which `allreduce()` calls first."""

    def __init__(self, parameters):
        params = [p for p in parameters if p.requires_grad]
        if not params:
            raise ValueError("GradArena: no trainable parameters")
        ids = set(map(id, params))
        self.stores, covered = [], set()
        for p in params:
            st = getattr(p, "_dpf_flat", None)
            if st is not None and id(st) not in covered and st.attached() and all(id(q) in ids for q in st.params):
                covered.add(id(st))
                self.stores.append(st)
        in_store = set(id(q) for st in self.stores for q in st.params)
        self.others = [p for p in params if id(p) not in in_store]
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise ValueError("GradArena: all parameters must be float32 on one device")
        # every store's slice starts on a 256-byte boundary (the fused Adam step and the kernels' 16-byte accesses want their
        # buffers aligned); the few padding words are zeros and travel with the message
        def up(n):
            return (n + 63) // 64 * 64
        n_store = 0
        for st in self.stores:
            n_store = up(n_store) + st.flat_g.numel()
        total = up(n_store) + sum(p.numel() for p in self.others)
        self.buf = torch.zeros(total, dtype=torch.float32, device=dev)
        self.n_store = up(n_store)
        off = 0
        for st in self.stores:
            off = up(off)
            n = st.flat_g.numel()
            st.rebase_grads(self.buf[off:off + n])
            off += n
        off = up(off)
        self.views = []
        for p in self.others:
            v = self.buf[off:off + p.numel()].view(p.shape)
            if p.grad is not None:
                with torch.no_grad():
                    v.copy_(p.grad)
            p.grad = v
            self.views.append(v)
            off += p.numel()
        self.n_params = sum(p.numel() for p in params)

    def nbytes(self):
        return self.buf.numel() * 4

    def sync(self, keep_none=False):
        """Every gradient of the model is in `buf` afterwards (missing ones as zeros), every .grad a view of it again.
        Fresh gradients (autograd hands a parameter whose .grad is None its gradient tensor without a kernel) are moved into
        the message with ONE multi-tensor copy.  keep_none (what `allreduce()` passes in a single-process run): a parameter
        that took no part in the backward keeps `.grad is None` -- its slice of the message is zeroed all the same -- so that
        the optimizer skips it as the reference's does (`if p.grad is None: continue`, optimizers.py:40-41): no moment
        decay, no weight decay, no step count.  With more than one rank such a parameter receives the other ranks' average
        (zeros if nobody used it) and is stepped -- the DistributedDataParallel behaviour; that difference is inherent to
        a fixed-layout message."""
        for st in self.stores:
            st.attach_grads()
        src, dst, zero, none = [], [], [], set()
        for p, v in zip(self.others, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                zero.append(v)
                if keep_none:
                    none.add(id(p))
            else:
                src.append(p.grad); dst.append(v)
        with torch.no_grad():
            if zero:
                torch._foreach_zero_(zero)
            if dst:
                torch._foreach_copy_(dst, src)
        for p, v in zip(self.others, self.views):
            if id(p) not in none:
                p.grad = v

    def zero_grad(self):
        """One fill for the stores (their gradient views stay attached; they count as "no gradient yet"); the other
        parameters' gradients are dropped (set_to_none): autograd then hands each its gradient without an add kernel and
        `sync()` gathers them with one multi-tensor copy -- 30 in-place adds less per step for the autoencoder."""
        for st in self.stores:
            st.attach_grads()
            st.grad_written = False
        if self.n_store:
            self.buf[:self.n_store].zero_()
        for p in self.others:
            p.grad = None

    def allreduce(self, average=True):
        """THE collective of the step.  Returns the number of elements reduced (0 in a single-process run)."""
        rank, w = world()
        self.sync(keep_none=(w == 1))
        if w == 1:
            return 0
        for st in self.stores:
            st.grad_written = True
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        if average:
            self.buf.div_(w)
        return self.buf.numel()


class count_collectives:
    """Context manager that COUNTS the collectives issued through torch.distributed while it is active (all_reduce,
    all_gather[_into_tensor], reduce_scatter[_tensor], broadcast, reduce, all_to_all[_single]; barriers are not
    data-path collectives and are listed separately).  bench.py reports `collectives_per_step` from it and
    tests/dist_worker_gpu.py asserts the north star's "a single all-reduce per step" with it."""
    NAMES = ("all_reduce", "all_gather", "all_gather_into_tensor", "reduce_scatter", "reduce_scatter_tensor", "broadcast",
             "reduce", "all_to_all", "all_to_all_single")

    def __init__(self):
        self.calls = {}
        self.barriers = 0
        self._saved = {}

    def total(self):
        return sum(self.calls.values())

    def __enter__(self):
        def wrap(name, fn):
            def counted(*a, **k):
                self.calls[name] = self.calls.get(name, 0) + 1
                return fn(*a, **k)
            return counted
        for name in self.NAMES:
            fn = getattr(dist, name, None)
            if fn is not None:
                self._saved[name] = fn
                setattr(dist, name, wrap(name, fn))
        bar = dist.barrier

        def counted_barrier(*a, **k):
            self.barriers += 1
            return bar(*a, **k)
        self._saved["barrier"] = bar
        dist.barrier = counted_barrier
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        self._saved = {}
        return False


def broadcast_buffers(module, src=0):
    """BatchNorm running statistics diverge across replicas (no SyncBN); broadcast rank `src`'s
    before a checkpoint so that every rank saves the same state dict."""
    rank, w = world()
    if w == 1:
        return
    for b in module.buffers():
        if b.dtype.is_floating_point:
            dist.broadcast(b, src)
