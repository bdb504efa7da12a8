// StructuralLossesBackend as a COMPILED torch extension over the C ABI of libdpf_hip.so -- what a C++ / pybind caller of the
// reference links instead of its CUDA extension.  Same five entry points, argument order, shapes, dtypes and error behaviour
// as lib/metrics/pytorch_structural_losses/pybind/bind.cpp:9-15 + src/structural_loss.cpp:18-139 (every tensor must be a
// contiguous float32 -- indices int32 -- tensor on the GPU; outputs are allocated here; the work runs on the current stream).
// The Python module of the same name (metrics/StructuralLosses/StructuralLossesBackend.py, ctypes) stays the default host
// side; this file exists so that nothing of the boundary is Python-only.  No kernels here: plain C++ against include/dpf_hip.h.
// (PyTorch-ROCm presents its HIP devices as device type "cuda": the guard and stream classes to use are the "masquerading" ones)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <initializer_list>
#include <vector>

#include "dpf_hip.h"

namespace {

using at::Tensor;

struct Shape3 { int64_t b, n, m; };

void need(const Tensor &t, const char *name, at::ScalarType dtype = at::kFloat) {
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");                       // structural_loss.cpp:10
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");                    // structural_loss.cpp:11
    TORCH_CHECK(t.scalar_type() == dtype, name, " must be ", dtype == at::kFloat ? "float32" : "int32");
}

Shape3 clouds(const Tensor &set_d, const Tensor &set_q) {
    need(set_d, "set_d");
    need(set_q, "set_q");
    TORCH_CHECK(set_d.dim() == 3 && set_q.dim() == 3 && set_d.size(2) == 3 && set_q.size(2) == 3 && set_d.size(0) == set_q.size(0),
                "expected set_d (B, n, 3) and set_q (B, m, 3)");
    return {set_d.size(0), set_d.size(1), set_q.size(1)};
}

Tensor fresh(const Tensor &like, std::initializer_list<int64_t> shape, at::ScalarType dtype = at::kFloat) {
    return at::empty(shape, like.options().dtype(dtype));
}

dpf_stream_t stream_of(const Tensor &t) { return (dpf_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void ok(int rc, const char *what) { TORCH_CHECK(rc == 0, "dpf_hip: ", what, " failed with code ", rc); }

}  // namespace

// structural_loss.cpp:27-40: match (B, m, n), temp (B, 2 (n + m))
std::vector<Tensor> ApproxMatch(Tensor set_d, Tensor set_q) {
    const Shape3 s = clouds(set_d, set_q);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(set_d.device());
    Tensor match = fresh(set_d, {s.b, s.m, s.n}), temp = fresh(set_d, {s.b, (s.n + s.m) * 2});
    // the write-once path wants scratch (include/dpf_hip.h); without it the call falls back to read-modify-write
    const size_t wsb = dpf_approxmatch_workspace_bytes((int)s.b, (int)s.n, (int)s.m);
    Tensor ws = fresh(set_d, {(int64_t)wsb}, at::kByte);
    ok(dpf_approxmatch_ws((int)s.b, (int)s.n, (int)s.m, set_d.data_ptr<float>(), set_q.data_ptr<float>(), match.data_ptr<float>(),
                          temp.data_ptr<float>(), ws.data_ptr(), wsb, stream_of(set_d)), "approxmatch");
    return {match, temp};
}

// structural_loss.cpp:42-55: cost (B,)
Tensor MatchCost(Tensor set_d, Tensor set_q, Tensor match) {
    const Shape3 s = clouds(set_d, set_q);
    need(match, "match");
    TORCH_CHECK(match.dim() == 3 && match.size(0) == s.b && match.size(1) == s.m && match.size(2) == s.n, "expected match (B, m, n)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(set_d.device());
    Tensor out = fresh(set_d, {s.b});
    ok(dpf_matchcost((int)s.b, (int)s.n, (int)s.m, set_d.data_ptr<float>(), set_q.data_ptr<float>(), match.data_ptr<float>(),
                     out.data_ptr<float>(), stream_of(set_d)), "matchcost");
    return out;
}

// structural_loss.cpp:57-72: grad1 (B, n, 3), grad2 (B, m, 3)
std::vector<Tensor> MatchCostGrad(Tensor set_d, Tensor set_q, Tensor match) {
    const Shape3 s = clouds(set_d, set_q);
    need(match, "match");
    TORCH_CHECK(match.dim() == 3 && match.size(0) == s.b && match.size(1) == s.m && match.size(2) == s.n, "expected match (B, m, n)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(set_d.device());
    Tensor g1 = fresh(set_d, {s.b, s.n, 3}), g2 = fresh(set_d, {s.b, s.m, 3});
    ok(dpf_matchcostgrad((int)s.b, (int)s.n, (int)s.m, set_d.data_ptr<float>(), set_q.data_ptr<float>(), match.data_ptr<float>(),
                         g1.data_ptr<float>(), g2.data_ptr<float>(), stream_of(set_d)), "matchcostgrad");
    return {g1, g2};
}

// structural_loss.cpp:83-103: dist1, idx1 (B, n); dist2, idx2 (B, m); idx int32
std::vector<Tensor> NNDistance(Tensor set_d, Tensor set_q) {
    const Shape3 s = clouds(set_d, set_q);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(set_d.device());
    Tensor d1 = fresh(set_d, {s.b, s.n}), i1 = fresh(set_d, {s.b, s.n}, at::kInt);
    Tensor d2 = fresh(set_d, {s.b, s.m}), i2 = fresh(set_d, {s.b, s.m}, at::kInt);
    ok(dpf_nndistance_auto((int)s.b, (int)s.n, set_d.data_ptr<float>(), (int)s.m, set_q.data_ptr<float>(), d1.data_ptr<float>(),
                           i1.data_ptr<int>(), d2.data_ptr<float>(), i2.data_ptr<int>(), stream_of(set_d)), "nndistance");
    return {d1, i1, d2, i2};
}

// structural_loss.cpp:105-139: grad1 (B, n, 3), grad2 (B, m, 3)
std::vector<Tensor> NNDistanceGrad(Tensor set_d, Tensor set_q, Tensor idx1, Tensor idx2, Tensor grad_dist1, Tensor grad_dist2) {
    const Shape3 s = clouds(set_d, set_q);
    need(idx1, "idx1", at::kInt); need(idx2, "idx2", at::kInt);
    need(grad_dist1, "grad_dist1"); need(grad_dist2, "grad_dist2");
    TORCH_CHECK(idx1.numel() == s.b * s.n && grad_dist1.numel() == s.b * s.n && idx2.numel() == s.b * s.m && grad_dist2.numel() == s.b * s.m,
                "expected idx1 / grad_dist1 (B, n) and idx2 / grad_dist2 (B, m)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(set_d.device());
    Tensor g1 = fresh(set_d, {s.b, s.n, 3}), g2 = fresh(set_d, {s.b, s.m, 3});
    ok(dpf_nndistancegrad((int)s.b, (int)s.n, set_d.data_ptr<float>(), (int)s.m, set_q.data_ptr<float>(), grad_dist1.data_ptr<float>(),
                          idx1.data_ptr<int>(), grad_dist2.data_ptr<float>(), idx2.data_ptr<int>(), g1.data_ptr<float>(),
                          g2.data_ptr<float>(), stream_of(set_d)), "nndistancegrad");
    return {g1, g2};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {        // the names of pybind/bind.cpp:9-15
    m.def("ApproxMatch", &ApproxMatch);
    m.def("MatchCost", &MatchCost);
    m.def("MatchCostGrad", &MatchCostGrad);
    m.def("NNDistance", &NNDistance);
    m.def("NNDistanceGrad", &NNDistanceGrad);
}
