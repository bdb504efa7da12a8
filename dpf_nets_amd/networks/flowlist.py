"""Lazy list of per-layer (B,3,N) tensors backed by one (L,B,3,N) buffer.

LocalCondRNVPDecoder.forward must return three python-list-like objects of
3*n_flows tensors in DIRECT order (lib/networks/decoders.py:54-72) which the
callers index ([0], [-1]), concatenate with `+` / `+=` (lib/networks/models.py:
119-122,169-171,214-216) and pass to sum() (lib/networks/losses.py:13).  The
fused HIP stack writes each list into a single buffer; this sequence hands out
views on demand instead of materialising 189 tensor objects per call, and
remembers the layer-sum the kernel accumulated in registers."""
from collections.abc import Sequence


class FlowList(Sequence):
    def __init__(self, buf, total=None):
        self._buf = buf            # (L, B, 3, N)
        self._total = total        # (B, 3, N) = sum over L from the kernel, or None
        self._views = None
        self._token = object()

    def __len__(self):
        return self._buf.shape[0]

    def views(self):
        if self._views is None:
            self._views = list(self._buf.unbind(0))
            # tag the views so that a python list built by `[prior] + flowlist` can still be
            # recognised by losses.total_logvar (no reference back to self: no cycles)
            for i, v in enumerate(self._views):
                v._dpf_pos = (self._token, i)
            if self._total is not None and self._views:
                self._views[-1]._dpf_total = (self._token, len(self._views), self._total)
        return self._views

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.views()[i]
        if self._views is not None:
            return self._views[i]
        L = len(self)
        if not -L <= i < L:
            raise IndexError(i)
        return self._buf[i]

    def __iter__(self):
        return iter(self.views())

    def __add__(self, other):
        return self.views() + list(other)

    def __radd__(self, other):
        return list(other) + self.views()

    def total(self):
        if self._total is None:
            self._total = self._buf.sum(0)
        return self._total

    @property
    def stacked(self):
        return self._buf
