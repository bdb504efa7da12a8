"""Adam / LRUpdater with the reference's signatures, hyper-parameters, arithmetic and state layout
(lib/networks/optimizers.py:8-98), so `train_ae.py:63-64` and resumed optimizer checkpoints work unchanged.

The reference's `step()` is a Python loop over the parameters with ~12 tensor ops each: >20 000 kernel launches per
step for the 2 000+ parameter tensors of an n_flows=21 model -- more wall time than the whole forward/backward once the
decoder runs in HIP.  Here the SAME operations, in the same order, run

  * over the flat buffers of a decoder whose parameters live in a FlatStore (`flatten_parameters()`): one sequence of
    ~12 ops for all of its 32*L parameters (elementwise => bitwise the per-parameter results; the per-parameter state
    entries are views of flat state buffers, so `state_dict()` keeps the reference's layout);
  * as multi-tensor (`torch._foreach_*`) ops over everything else.

SURVEY 8(f) rank 4 asks for a fused AMSGrad-Adam step: on CUDA tensors the flat-store update is ONE launch of
`dpf_adam_step` (csrc/adam.hip; r03) -- the same operations with the same roundings as the op sequence, bit for bit
(tests/test_gpu_adam.py), 36 B per parameter in one pass instead of ~12 passes; `DPF_FUSED_ADAM=0` keeps the op sequence.
Everything else (CPU tensors, parameters outside a store) runs the op sequence / the multi-tensor ops.
"""
import math
import os

import numpy as np
import torch
from torch.optim import Optimizer

FUSED_ADAM = os.environ.get("DPF_FUSED_ADAM", "1") != "0"


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        self._flat = {}          # id(store) -> flat state buffers
        self._fast = {}          # id(param group) -> (its whole FlatStores, its other parameters, len, first, last)

    # ---- the update of optimizers.py:52-74 on lists of tensors (lists of one flat tensor for a FlatStore)
    @staticmethod
    def _update_flat(p, grad, exp_avg, exp_avg_sq, max_sq, step, lr, beta1, beta2, eps, weight_decay, amsgrad):
        """The update of ONE flat fp32 CUDA buffer: the fused HIP kernel when it applies (-> True), else False."""
        if not (FUSED_ADAM and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and grad.is_contiguous()
                and p.data_ptr() % 16 == 0 and grad.data_ptr() % 16 == 0):
            return False
        bc1, bc2 = 1 - beta1 ** step, math.sqrt(1 - beta2 ** step)
        if bc1 == 0.0 or bc2 == 0.0:
            return False
        from .._lib import lib, check, current_stream
        with torch.cuda.device(p.device):
            check(lib().dpf_adam_step(p.numel(), p.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                      max_sq.data_ptr() if amsgrad else None, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                                      current_stream()), "adam_step")
        return True

    @staticmethod
    def _update(ps, grads, exp_avgs, exp_avg_sqs, max_sqs, step, lr, beta1, beta2, eps, weight_decay, amsgrad):
        torch._foreach_mul_(exp_avgs, beta1)
        torch._foreach_add_(exp_avgs, grads, alpha=1 - beta1)                        # :53
        torch._foreach_mul_(exp_avg_sqs, beta2)
        torch._foreach_addcmul_(exp_avg_sqs, grads, grads, value=1 - beta2)          # :54
        if amsgrad:
            torch._foreach_maximum_(max_sqs, exp_avg_sqs)                            # :57
            denom = torch._foreach_sqrt(max_sqs)
        else:
            denom = torch._foreach_sqrt(exp_avg_sqs)
        bc1 = 1 - beta1 ** step
        bc2 = math.sqrt(1 - beta2 ** step)
        exp_avg_c = torch._foreach_div(exp_avgs, bc1)                                # :66
        torch._foreach_div_(denom, bc2)
        torch._foreach_add_(denom, eps)                                              # :67
        if weight_decay != 0:                                                        # :69-72
            upd = torch._foreach_mul(ps, weight_decay)
            torch._foreach_addcdiv_(upd, exp_avg_c, denom, value=lr)
            torch._foreach_sub_(ps, upd)
        else:
            torch._foreach_addcdiv_(ps, exp_avg_c, denom, value=-lr)                 # :74

    def _init_state(self, p, amsgrad):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p.data)
            st["exp_avg_sq"] = torch.zeros_like(p.data)
            if amsgrad:
                st["max_exp_avg_sq"] = torch.zeros_like(p.data)
        return st

    def _flat_state(self, store, amsgrad):
        """Flat moment buffers of a FlatStore; the per-parameter state entries are views of them.  Rebuilt (values
        carried over) when the aliasing is gone, e.g. after load_state_dict, which copies every state tensor."""
        names = ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ())
        fs = self._flat.get(id(store))
        first, last = store.params[0], store.params[-1]

        def aliased(f):
            sa, sb = self.state.get(first, {}), self.state.get(last, {})
            return all(n in sa and n in sb and sa[n].data_ptr() == f[n][0].data_ptr() and sb[n].data_ptr() == f[n][-1].data_ptr()
                       for n in names)
        if fs is not None and fs["flat_p"] is store.flat_p and aliased(fs["views"]):
            return fs
        fs = {"flat_p": store.flat_p, "buf": {}, "views": {}}
        for n in names:
            buf = torch.zeros_like(store.flat_p)
            views = [buf[v.data_ptr() // 4 - store.flat_p.data_ptr() // 4:][:v.numel()].view(v.shape) for v in store.pviews]
            old = [self.state[p].get(n) if p in self.state else None for p in store.params]
            if any(o is not None for o in old):
                with torch.no_grad():
                    torch._foreach_copy_([v for v, o in zip(views, old) if o is not None], [o for o in old if o is not None])
            fs["buf"][n], fs["views"][n] = buf, views
        for i, p in enumerate(store.params):
            st = self.state[p]
            st.setdefault("step", 0)
            for n in names:
                st[n] = fs["views"][n][i]
        fs["pstates"] = [self.state[p] for p in store.params]       # the per-parameter dicts, without hashing a tensor each step
        self._flat[id(store)] = fs
        return fs

    def _split_group(self, group):
        """(FlatStores whose parameters all belong to this group, the group's other parameters) -- cached per group, keyed
        by the list's length and end points."""
        params = group["params"]
        c = self._fast.get(id(group))
        # stale when the list changed, or when a parameter's store was rebuilt (module.to() / .cuda() re-assign .data and
        # the decoder builds a new FlatStore: the cached one would pin the slow path for good)
        if c is None or c[2] != len(params) or (len(params) and (
                c[3] is not params[0] or c[4] is not params[-1] or
                c[5] is not getattr(params[0], "_dpf_flat", None) or c[6] is not getattr(params[-1], "_dpf_flat", None))):
            ids = set(map(id, params))
            stores, covered = [], set()
            for p in params:
                st = getattr(p, "_dpf_flat", None)
                if st is not None and id(st) not in covered and all(id(q) in ids for q in st.params):
                    covered.add(id(st))
                    stores.append(st)
            in_store = set(id(q) for st in stores for q in st.params)
            rest = [p for p in params if id(p) not in in_store]
            c = self._fast[id(group)] = (stores, rest, len(params), params[0] if len(params) else None,
                                         params[-1] if len(params) else None,
                                         getattr(params[0], "_dpf_flat", None) if len(params) else None,
                                         getattr(params[-1], "_dpf_flat", None) if len(params) else None)
        return c[0], c[1]

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad, except for the FlatStores a group holds whole: their gradients are persistent
        views of ONE buffer, so they are cleared by one fill and stay attached, whatever `set_to_none` says -- the state the
        next backward would re-create anyway (FlatStore.attach_grads), and what `optimizer.zero_grad()` meant when the
        reference was written (PyTorch < 2.0 zeroed in place); walking the 2016 parameters of n_flows = 21 twice per step
        (here to drop the views, in the backward to put them back) costs ~1.2 ms of host time.  The meaning of
        `set_to_none` is kept through the store's `grad_written` flag: with set_to_none=True (the default) a step() that
        no backward (FlatStore.accumulate), gradient exchange or attach of a foreign gradient preceded SKIPS the store,
        as stock PyTorch skips parameters whose .grad is None (no moment decay, no weight decay, no step count); with
        set_to_none=False the zeros count as gradients, as they do there."""
        for group in self.param_groups:
            stores, rest = self._split_group(group)
            others = list(rest)
            for store in stores:
                ps, gv, mid = store.params, store.gviews, len(store.params) // 2
                if store.attached() and ps[0].grad is gv[0] and ps[mid].grad is gv[mid] and ps[-1].grad is gv[-1]:
                    store.flat_g.zero_()
                    store.grad_written = not set_to_none
                else:
                    others += store.params
            for p in others:
                if p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    if p.grad.grad_fn is not None:
                        p.grad.detach_()
                    else:
                        p.grad.requires_grad_(False)
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (beta1, beta2), eps = group["lr"], group["betas"], group["eps"]
            wd, amsgrad = group["weight_decay"], group["amsgrad"]
            # ---- the common case of a training loop over a flattened decoder: the group holds whole FlatStores with every
            # gradient attached -- a handful of checks and one sequence of ops on the flat buffers per store, no
            # per-parameter pass but the step counters (the general path below walks the 2016 parameters of n_flows = 21
            # about ten times); the group's other parameters, and any store that fails a check, take the general path
            stores, rest = self._split_group(group)
            general = list(rest)
            for store in stores:
                done = False
                if store.attached():
                    ps, gv, mid = store.params, store.gviews, len(store.params) // 2
                    if ps[0].grad is gv[0] and ps[mid].grad is gv[mid] and ps[-1].grad is gv[-1]:
                        if not getattr(store, "grad_written", True):           # zero_grad(set_to_none=True) and nothing since: "grad is None"
                            continue
                        fs = self._flat_state(store, amsgrad)
                        pst = fs["pstates"]
                        if pst[0]["step"] == pst[mid]["step"] == pst[-1]["step"]:
                            step = pst[0]["step"] + 1
                            for st in pst:
                                st["step"] = step
                            if not self._update_flat(store.flat_p, store.flat_g, fs["buf"]["exp_avg"], fs["buf"]["exp_avg_sq"],
                                                     fs["buf"]["max_exp_avg_sq"] if amsgrad else None, step, lr, beta1, beta2, eps, wd, amsgrad):
                                self._update([store.flat_p], [store.flat_g], [fs["buf"]["exp_avg"]], [fs["buf"]["exp_avg_sq"]],
                                             [fs["buf"]["max_exp_avg_sq"]] if amsgrad else None, step, lr, beta1, beta2, eps, wd, amsgrad)
                            done = True
                if not done:
                    general += store.params
            if not general:
                continue
            todo = [p for p in general if p.grad is not None]
            if any(p.grad.is_sparse for p in todo):
                raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
            in_group = set(id(p) for p in todo)
            # ---- parameters that live in a flat store, one sequence of ops per store
            stores, seen = [], set()
            for p in todo:
                store = getattr(p, "_dpf_flat", None)
                if store is None or id(store) in seen:
                    continue
                seen.add(id(store))
                steps = set(self.state[q].get("step", 0) if q in self.state else 0 for q in store.params)
                if store.attached() and all(id(q) in in_group for q in store.params) and len(steps) == 1 and \
                        all(q.grad is g for q, g in zip(store.params, store.gviews)):
                    stores.append(store)
            flat_ids = set()
            for store in stores:
                fs = self._flat_state(store, amsgrad)
                step = self.state[store.params[0]]["step"] + 1
                for q in store.params:
                    self.state[q]["step"] = step
                    flat_ids.add(id(q))
                if not self._update_flat(store.flat_p, store.flat_g, fs["buf"]["exp_avg"], fs["buf"]["exp_avg_sq"],
                                         fs["buf"]["max_exp_avg_sq"] if amsgrad else None, step, lr, beta1, beta2, eps, wd, amsgrad):
                    self._update([store.flat_p], [store.flat_g], [fs["buf"]["exp_avg"]], [fs["buf"]["exp_avg_sq"]],
                                 [fs["buf"]["max_exp_avg_sq"]] if amsgrad else None, step, lr, beta1, beta2, eps, wd, amsgrad)
            # ---- everything else, multi-tensor, grouped by step count (and device / dtype)
            buckets = {}
            for p in todo:
                if id(p) in flat_ids:
                    continue
                st = self._init_state(p, amsgrad)
                st["step"] += 1
                buckets.setdefault((st["step"], p.device, p.dtype), []).append(p)
            for (step, _, _), ps in buckets.items():
                sts = [self.state[p] for p in ps]
                self._update([p.data for p in ps], [p.grad.data for p in ps], [s["exp_avg"] for s in sts],
                             [s["exp_avg_sq"] for s in sts], [s["max_exp_avg_sq"] for s in sts] if amsgrad else None,
                             step, lr, beta1, beta2, eps, wd, amsgrad)
        return loss


class LRUpdater(object):
    """Cosine schedule of the learning rate and of beta2 over `cycle_length` epochs (optimizers.py:78-98)."""

    def __init__(self, epoch_length, **kwargs):
        self.epoch_length = epoch_length
        self.cycle_length = kwargs["cycle_length"]
        self.min_lr, self.max_lr = kwargs["min_lr"], kwargs["max_lr"]
        self.beta1 = kwargs["beta1"]
        self.min_beta2, self.max_beta2 = kwargs["min_beta2"], kwargs["max_beta2"]

    def __call__(self, optimizer, epoch, iteration):
        rel_epoch = epoch % self.cycle_length
        cur_step = (rel_epoch * self.epoch_length + iteration) / (self.cycle_length * self.epoch_length)
        cur_lr = self.min_lr + 0.5 * (self.max_lr - self.min_lr) * (1.0 + np.cos(np.pi * cur_step))
        cur_beta2 = self.min_beta2 + 0.5 * (self.max_beta2 - self.min_beta2) * (1.0 + np.cos(np.pi * cur_step))
        for group in optimizer.param_groups:
            group["lr"] = cur_lr
            group["betas"] = (self.beta1, cur_beta2)
