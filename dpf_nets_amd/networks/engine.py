"""Host driver of the fused HIP flow stack (csrc/flow.hip via include/dpf_hip.h).

A FlowStack owns, for an ordered list of coupling layers (DIRECT order,
lib/networks/decoders.py:58-64):
  * the canonical fp32 parameter block the C ABI consumes (dpf_hip.h),
  * the packed MFMA-fragment weights per precision (built by dpf_flow_pack,
    once per weight version -- never per batch),
and runs dpf_flow_film + dpf_flow_forward on the current stream.  torch is used
for device buffers and the stream handle only."""
import os

import torch

from .._lib import lib, check, current_stream, PREC, MODE

# "f16x3" (default): W1 and the hidden activations as fp16 hi + lo parts, three MFMA products -- ~5e-7 of the reference
# (fp32-class), and the cheapest split on the VALU; "bf16x3": the same with bf16 parts (~7e-6; no fp16 range limit);
# "bf16x6": three bf16 parts, six products (fp32-class, no range limit); "bf16": one product (~6e-4, speed reference only)
DEFAULT_PRECISION = os.environ.get("DPF_PRECISION", "f16x3")
F = 64
# f16x3's hi/lo split of the hidden activations is exact while |h0| < 2048 (the lo part is clamped to [0, 1]: above, the
# split degrades to fp16-hi precision, and at 65504 it saturates -- csrc/flow_common.h split_relu_f16).  At pack time (once
# per weight version) the BN0-folded first layer gives a bound on |h0| for coordinates up to F16_COORD_MAX; a checkpoint
# whose bound reaches F16_LIMIT (e.g. a collapsed running variance) evaluates at bf16x6 instead, with a warning.
F16_LIMIT = 2048.0
F16_COORD_MAX = 32.0


def _pad_cols(w, n):
    if w.shape[1] == n:
        return w
    return torch.cat([w, w.new_zeros(w.shape[0], n - w.shape[1])], dim=1)


def _pad_rows(w, n):
    if w.shape[0] == n:
        return w
    return torch.cat([w, w.new_zeros(n - w.shape[0], w.shape[1])], dim=0)


def layer_canon_pieces(layer):
    """Flattened tensors of one CondRealNVPFlow3D in the canonical order of dpf_hip.h."""
    out = []
    for br in ("logvar", "mu"):
        t0 = getattr(layer, "T_%s_0" % br)
        sd0, bn0, sd1, bn1 = t0[0], t0[1], t0[3], t0[4]
        sd2 = getattr(layer, "T_%s_1" % br)[1]
        out += [_pad_cols(sd0.weight[0], 2), bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var,
                sd1.weight[0], bn1.running_mean, bn1.running_var,
                _pad_rows(sd2.weight[0], 2), torch.cat([sd2.bias[0], sd2.bias.new_zeros(4 - sd2.bias.shape[1])])]
        for s in ("w", "b"):
            tc = getattr(layer, "T_%s_0_cond_%s" % (br, s))
            lin0, bnf, lin1 = tc[0], tc[1], tc[3]
            out += [lin0.weight, bnf.weight, bnf.bias, bnf.running_mean, bnf.running_var, lin1.weight, lin1.bias]
    return [t.detach().reshape(-1) for t in out]


def layer_meta(layer):
    k = list(layer.keep_inds) + [-1] * (2 - len(layer.keep_inds))
    w = list(layer.warp_inds) + [-1] * (2 - len(layer.warp_inds))
    return k + w


class FlowStack:
    def __init__(self, layers):
        self.layers = list(layers)
        self._canon = None
        self._meta = None
        self._packed = {}
        self._sentinels = []
        self._sentinel_state = None
        self._f16_bound = None
        self.last_precision = None
        for lyr in (self.layers[0], self.layers[-1]):
            self._sentinels += [lyr.T_mu_0[3].weight, lyr.T_logvar_1[1].weight]

    # -- weight-version tracking -------------------------------------------------
    def invalidate(self):
        self._canon = None
        self._packed = {}

    def _sentinel(self):
        return tuple((t._version, t.data_ptr()) for t in self._sentinels)

    def _check_fresh(self):
        s = self._sentinel()
        if s != self._sentinel_state:
            self.invalidate()
            self._sentinel_state = s

    # -- packing -----------------------------------------------------------------
    def _ensure(self, precision, device, L=None):
        """canon block, meta and the packed weights of the first L layers (default: all)."""
        self._check_fresh()
        nl = len(self.layers)
        L = nl if L is None else L
        lyr0 = self.layers[0]
        if lyr0.f_n_features != F:
            raise RuntimeError("dpf_hip flow kernels are built for f_n_features == 64, got %d" % lyr0.f_n_features)
        G = lyr0.g_n_features
        if self._canon is None or self._canon.device != device:
            pieces = []
            for lyr in self.layers:
                pieces += layer_canon_pieces(lyr)
            canon = torch.cat(pieces).to(device=device, dtype=torch.float32).contiguous()
            assert canon.numel() == nl * lib().dpf_flow_canon_floats(G), (canon.numel(), nl, G)
            self._canon = canon
            self._meta = torch.tensor([layer_meta(l) for l in self.layers], dtype=torch.int32, device=device)
            self._packed = {}
            self._f16_bound = None
        if precision == "f16x3" and not self.f16_in_range(device):
            precision = "bf16x6"               # fp32-class without a range limit (slower: six products, unpipelined body)
        key = (precision, L)
        if key not in self._packed:
            nbytes = lib().dpf_flow_packed_bytes(L, G, PREC[precision])
            packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(lib().dpf_flow_pack(L, G, PREC[precision], self._canon.data_ptr(), self._meta.data_ptr(),
                                      packed.data_ptr(), current_stream()), "flow_pack")
            self._packed[key] = packed
        return self._canon, self._meta, self._packed[key], G, precision

    def f16_in_range(self, device):
        """Once per weight version: max over layers / branches / features of  sum_k |s0 W0[f][k]| * F16_COORD_MAX + |T_f|
        (s0 = gamma0 / sqrt(running_var0 + eps), T = beta0 - running_mean0 * s0) -- the largest |BN0(W0 x)| a point with
        coordinates up to F16_COORD_MAX can produce -- and max |W1| against the fp16 range."""
        if self._f16_bound is None:
            with torch.no_grad():
                hb, wb = [], []
                for lyr in self.layers:
                    for br in ("logvar", "mu"):
                        t0 = getattr(lyr, "T_%s_0" % br)
                        sd0, bn0, sd1 = t0[0], t0[1], t0[3]
                        s0 = bn0.weight.detach().float() / torch.sqrt(bn0.running_var.detach().float() + bn0.eps)
                        w = sd0.weight.detach()[0].float().abs().sum(1) * s0.abs()
                        T = (bn0.bias.detach().float() - bn0.running_mean.detach().float() * s0).abs()
                        hb.append((w * F16_COORD_MAX + T).amax())
                        wb.append(sd1.weight.detach().float().abs().amax())
                both = torch.stack([torch.stack(hb).amax(), torch.stack(wb).amax()]).to("cpu")
            h0_bound, w1_max = float(both[0]), float(both[1])
            ok = (h0_bound < F16_LIMIT) and (w1_max < 65504.0)              # NaN compares False
            if not ok:
                import warnings
                warnings.warn("dpf_nets_amd: f16x3 is outside its exact range for these weights (|h0| bound %.3g, limit %g; "
                              "max|W1| %.3g): evaluating at bf16x6" % (h0_bound, F16_LIMIT, w1_max))
            self._f16_bound = (h0_bound, w1_max, ok)
        return self._f16_bound[2]

    # -- run ---------------------------------------------------------------------
    def run(self, p, g, mode, precision=None, want_lists=True, n_layers=None, want_pointmajor=False, base=None):
        """p (B,3,N), g (B,G) fp32 CUDA.  Returns (p_out, sum_logvar, ps, mus, logvars); the
        last three are (L,B,3,N) buffers in DIRECT order or None.  With want_pointmajor the
        (B,N,3) copy of p_out is left in self.last_pointmajor.
        base = (mu0, lv0) (direct mode only): p is the NOISE and the stack starts from
        z = p * exp(0.5 * lv0) + mu0 (models.py:76-79), formed in the kernel's prologue from the (possibly stride-0
        expanded) (B,3,N) views mu0 / lv0; z is left in self.last_base_sample."""
        precision = precision or DEFAULT_PRECISION
        if precision not in PREC:
            raise ValueError("precision must be one of %s" % sorted(PREC))
        if not p.is_cuda or not g.is_cuda:
            raise RuntimeError("the fused flow stack runs on MI355X only (p and g must be CUDA tensors); "
                               "there is no CPU fallback")
        if p.dtype != torch.float32 or g.dtype != torch.float32:
            raise RuntimeError("p and g must be float32")
        if p.dim() != 3 or p.shape[1] != 3 or g.dim() != 2 or g.shape[0] != p.shape[0]:
            raise RuntimeError("expected p (B,3,N) and g (B,G)")
        p = p.contiguous()
        g = g.contiguous()
        B, _, N = p.shape
        with torch.cuda.device(p.device):
            L = len(self.layers) if n_layers is None else int(n_layers)
            if not 0 < L <= len(self.layers):
                raise ValueError("n_layers out of range")
            canon, meta, packed, G, precision = self._ensure(precision, p.device, L)
            self.last_precision = precision
            if g.shape[1] != G:
                raise RuntimeError("g has %d features, the layers expect %d" % (g.shape[1], G))
            dev = p.device
            film = torch.empty(lib().dpf_flow_film_floats(L, B), dtype=torch.float32, device=dev)
            p_out = torch.empty_like(p)
            sum_lv = torch.empty_like(p)
            pm = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if want_pointmajor else None
            self.last_pointmajor = pm
            if want_lists:
                lists = torch.empty((3, L, B, 3, N), dtype=torch.float32, device=dev)
                lp = [lists[i].data_ptr() for i in range(3)]
            else:
                lists, lp = None, [None, None, None]
            eps = float(self.layers[0].eps_value)
            stream = current_stream()
            check(lib().dpf_flow_film(L, B, G, PREC[precision], packed.data_ptr(), g.data_ptr(), film.data_ptr(), eps,
                                      stream), "flow_film")
            self.last_base_sample = None
            if base is not None:
                if mode != "direct":
                    raise ValueError("a base distribution is sampled in direct mode only")
                mu0, lv0 = base
                for t in (mu0, lv0):
                    if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (B, 3, N):
                        raise RuntimeError("base mu / logvar must be float32 CUDA tensors (views) of shape (B,3,N)")
                z = torch.empty_like(p)
                self.last_base_sample = z
                check(lib().dpf_flow_forward_base(L, B, N, PREC[precision], packed.data_ptr(), meta.data_ptr(), film.data_ptr(),
                                                  p.data_ptr(), mu0.data_ptr(), *mu0.stride(), lv0.data_ptr(), *lv0.stride(),
                                                  z.data_ptr(), p_out.data_ptr(), pm.data_ptr() if pm is not None else None,
                                                  sum_lv.data_ptr(), lp[0], lp[1], lp[2], eps, stream), "flow_forward_base")
            else:
                check(lib().dpf_flow_forward(L, B, N, MODE[mode], PREC[precision], packed.data_ptr(), meta.data_ptr(),
                                             film.data_ptr(), p.data_ptr(), p_out.data_ptr(),
                                             pm.data_ptr() if pm is not None else None, sum_lv.data_ptr(),
                                             lp[0], lp[1], lp[2], eps, stream), "flow_forward")
        if want_lists:
            return p_out, sum_lv, lists[0], lists[1], lists[2]
        return p_out, sum_lv, None, None, None
