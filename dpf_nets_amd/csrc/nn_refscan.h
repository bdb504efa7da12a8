// The reference's candidate scan for ONE query, restated operation by operation (NmDistanceKernel,
// lib/metrics/pytorch_structural_losses/src/nndistance.cu:5-122): candidates in batches of 512; inside a batch the first
// candidate is taken unconditionally and a later one under a strict '<' (:26, 35, 44, 53, 116); across batches the running
// result is replaced under a strict '>' (:120).  For finite distances that is "the first global minimum", which the fast
// kernels compute by other means (csrc/chamfer.hip, csrc/chamfer_mfma.hip); for NON-FINITE input (NaN / Inf coordinates,
// or coordinates large enough for a squared distance to overflow) every comparison with a NaN is false and the batch
// structure decides the result: a NaN distance at candidate 0 latches (NaN, 0), at the first candidate of a later batch it
// makes that batch lose, anywhere else the candidate is skipped.  The fast kernels detect non-finite input (queries: the
// running minimum never leaves +inf; candidates: an exponent test that rides on loads they make anyway) and hand the
// affected queries to this function, so the library's contract is the reference's result for ANY input
// (oracle/structural_oracle.c restates the same loop; tests/test_gpu_chamfer.py::test_nonfinite_*).
#ifndef DPF_NN_REFSCAN_H
#define DPF_NN_REFSCAN_H

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace {

constexpr int NN_REF_BATCH = 512;          // nndistance.cu:2

__device__ __forceinline__ void nn_reference_scan(const float *__restrict__ c, int nc, float x1, float y1, float z1,
                                               float &res_out, int &idx_out) {
    float res = 0.f;
    int res_i = 0;
    for (int k2 = 0; k2 < nc; k2 += NN_REF_BATCH) {
        const int end_k = min(nc, k2 + NN_REF_BATCH) - k2;
        float best = 0.f;
        int best_i = 0;
        for (int k = 0; k < end_k; ++k) {
            const float *p = c + (size_t)(k2 + k) * 3;
            const float x2 = p[0] - x1, y2 = p[1] - y1, z2 = p[2] - z1;
            const float d = (x2 * x2 + y2 * y2) + z2 * z2;
            if (k == 0 || d < best) { best = d; best_i = k + k2; }
        }
        if (k2 == 0 || res > best) { res = best; res_i = best_i; }
    }
    res_out = res;
    idx_out = res_i;
}

// true for NaN, +-Inf and anything whose square would overflow the sums the kernels form
__device__ __forceinline__ bool nn_not_finite(float v) { return !(v <= 3.0e38f); }
// true for a query the LDS-staged kernels cannot serve: their padding candidates sit at (3e38, 3e38, 3e38) "so far away that
// the distance is +inf" -- which holds for every query whose coordinates are below ~1e19 in magnitude (the difference then
// squares to +inf) and fails for a finite query out there, next to the padding: such a query takes the reference's scan too
__device__ __forceinline__ bool nn_query_far(float x, float y, float z) {
    return !(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) <= 1.0e18f);
}

}  // namespace
#endif
