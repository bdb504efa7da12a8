// PointFlowNLL (lib/networks/losses.py:11-15) as one pass + a fixed-order finish:
//   0.5 * ( sum_{b,c,n} [ sum_lv + lv0 + (s0 - mu0)^2 / exp(lv0) ] / B + log(2 pi) * C * N )
// s0 = the cloud at the base of the flow, sum_lv = the flow's summed log-variances (the fused stack accumulates them in
// registers), mu0 / lv0 = the base distribution's parameters -- in the reference stride-0 expansions of (1,C,1) or (B,C,1)
// tensors (models.py:108-117), read here through their strides, never materialised.  Replaces ~7 elementwise /
// reduction launches over (B,C,N) tensors; deterministic (per-workgroup partial sums added in index order).
#include <hip/hip_runtime.h>

#include "dpf_hip.h"

namespace {

constexpr int T = 256, MAXWG = 256;

__global__ __launch_bounds__(T) void nll_partial_kernel(int B, int C, int N, const float *__restrict__ s0, const float *__restrict__ mu0,
                                                        long mu_sb, long mu_sc, long mu_sn, const float *__restrict__ lv0, long lv_sb,
                                                        long lv_sc, long lv_sn, const float *__restrict__ sum_lv,
                                                        float *__restrict__ partial) {
    __shared__ float red[T / 64];
    const long total = (long)B * C * N;
    float acc = 0.f;
    for (long e = (long)blockIdx.x * T + threadIdx.x; e < total; e += (long)gridDim.x * T) {
        const int n = (int)(e % N), c = (int)((e / N) % C), b = (int)(e / ((long)N * C));
        const float lv = lv0[b * lv_sb + c * lv_sc + n * lv_sn], d = s0[e] - mu0[b * mu_sb + c * mu_sc + n * mu_sn];
        acc += (sum_lv ? sum_lv[e] : 0.f) + lv + d * d / expf(lv);
    }
#pragma unroll
    for (int m = 32; m; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void nll_finish_kernel(int nwg, int B, float constant, const float *__restrict__ partial, float *__restrict__ out) {
    float s = 0.f;
    for (int i = 0; i < nwg; ++i) s += partial[i];
    out[0] = 0.5f * (s / B + constant);
}

// d(out)/d(s0), d(out)/d(sum_lv) and, on request, d/d(mu0), d/d(lv0) as FULL (B,C,N) tensors (autograd's expand-backward
// reduces them over the broadcast dimensions when the base distribution is a learned per-cloud vector):
//   out = 0.5 * (S / B + const),  S = sum [ sum_lv + lv0 + (s0 - mu0)^2 / e^{lv0} ]
//   d s0 = g (s0 - mu0) / e^{lv0} / B,  d sum_lv = g / (2B),  d mu0 = -d s0,  d lv0 = g (1 - (s0 - mu0)^2 / e^{lv0}) / (2B)
__global__ __launch_bounds__(T) void nll_backward_kernel(int B, int C, int N, const float *__restrict__ s0, const float *__restrict__ mu0,
                                                         long mu_sb, long mu_sc, long mu_sn, const float *__restrict__ lv0, long lv_sb,
                                                         long lv_sc, long lv_sn, const float *__restrict__ gout, float *__restrict__ ds0,
                                                         float *__restrict__ dsum, float *__restrict__ dmu, float *__restrict__ dlv) {
    const long total = (long)B * C * N;
    const float g = gout[0] / (float)B;
    for (long e = (long)blockIdx.x * T + threadIdx.x; e < total; e += (long)gridDim.x * T) {
        const int n = (int)(e % N), c = (int)((e / N) % C), b = (int)(e / ((long)N * C));
        const float lv = lv0[b * lv_sb + c * lv_sc + n * lv_sn], d = s0[e] - mu0[b * mu_sb + c * mu_sc + n * mu_sn];
        const float iv = 1.0f / expf(lv), gd = g * d * iv;
        if (ds0) ds0[e] = gd;
        if (dsum) dsum[e] = 0.5f * g;
        if (dmu) dmu[e] = -gd;
        if (dlv) dlv[e] = 0.5f * g * (1.0f - d * d * iv);
    }
}

}  // namespace

extern "C" int dpf_pointflow_nll_backward(int B, int C, int N, const float *s0, const float *mu0, long mu_sb, long mu_sc, long mu_sn,
                                          const float *lv0, long lv_sb, long lv_sc, long lv_sn, const float *grad_out, float *d_s0,
                                          float *d_sum_lv, float *d_mu0, float *d_lv0, dpf_stream_t stream) {
    if (B <= 0 || C <= 0 || N <= 0 || !s0 || !mu0 || !lv0 || !grad_out) return DPF_EINVAL;
    const long total = (long)B * C * N;
    const int nwg = (int)((total + 4 * T - 1) / (4 * T) < 1024 ? (total + 4 * T - 1) / (4 * T) : 1024);
    hipLaunchKernelGGL(nll_backward_kernel, dim3(nwg), dim3(T), 0, (hipStream_t)stream, B, C, N, s0, mu0, mu_sb, mu_sc, mu_sn, lv0, lv_sb,
                       lv_sc, lv_sn, grad_out, d_s0, d_sum_lv, d_mu0, d_lv0);
    return (int)hipGetLastError();
}

extern "C" size_t dpf_pointflow_nll_workspace_floats(void) { return MAXWG; }

extern "C" int dpf_pointflow_nll(int B, int C, int N, const float *s0, const float *mu0, long mu_sb, long mu_sc, long mu_sn,
                                 const float *lv0, long lv_sb, long lv_sc, long lv_sn, const float *sum_lv, float *workspace, float *out,
                                 dpf_stream_t stream) {
    if (B <= 0 || C <= 0 || N <= 0 || !s0 || !mu0 || !lv0 || !workspace || !out) return DPF_EINVAL;
    const long total = (long)B * C * N;
    const int nwg = (int)((total + 4 * T - 1) / (4 * T) < MAXWG ? (total + 4 * T - 1) / (4 * T) : MAXWG);
    hipLaunchKernelGGL(nll_partial_kernel, dim3(nwg), dim3(T), 0, (hipStream_t)stream, B, C, N, s0, mu0, mu_sb, mu_sc, mu_sn, lv0, lv_sb, lv_sc,
                       lv_sn, sum_lv, workspace);
    hipLaunchKernelGGL(nll_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, nwg, B, 1.8378770664093453f * C * N, workspace, out);
    return (int)hipGetLastError();
}
