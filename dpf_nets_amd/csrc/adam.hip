// Fused AMSGrad-Adam step over ONE flat fp32 buffer ("fused AMSGrad-Adam step").
//
// Replaces the per-parameter loop of the reference's optimizer (lib/networks/optimizers.py:52-74, ~12 tensor ops per
// parameter: >20 000 launches per step at n_flows = 21) and the ~12-pass op sequence the Python mirror runs over a
// FlatStore (networks/optimizers.py::_update: 10 multi-tensor launches, ~0.2 ms per step for the 3.7 M parameters of the
// decoder, ~0.7 ms for the all_scaled model) by ONE pass: p, g, exp_avg, exp_avg_sq, max_exp_avg_sq read once, four of
// them written once -- 36 B per parameter, HBM-bound.
//
// Arithmetic: the SAME operations in the SAME order as the reference's step(), each rounded to fp32 where the tensor-op
// sequence rounds (every ATen op writes fp32), and contracted / rewritten exactly where PyTorch-ROCm's own kernels do it
// -- fixed empirically, op by op, against the GPU (tests/diag/aten_rounding_probe.py): `a + alpha * b` and
// `a + value * (b / c)` are ONE fused multiply-add, `sqrt` and `b / c` are correctly rounded, and a division by a Python
// scalar is a multiplication by the reciprocal computed in DOUBLE on the host and rounded to fp32.  The file is compiled
// with -ffp-contract=off: every contraction below is explicit.  tests/test_gpu_adam.py holds this kernel bit for bit to
// networks/optimizers.py::_update (= the reference's step()) over several steps, with and without AMSGrad / weight decay.
//   exp_avg    = exp_avg * beta1 + (1 - beta1) * g                         :52-53
//   exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) * g * g                  :54
//   max_sq     = max(max_sq, exp_avg_sq); denom = sqrt(max_sq)             :57-59   (denom = sqrt(exp_avg_sq) without amsgrad)
//   exp_avg_c  = exp_avg / bc1;  denom_c = denom / bc2 + eps               :63-67   (bc1 = 1 - beta1^t, bc2 = sqrt(1 - beta2^t))
//   p -= p * wd + lr * exp_avg_c / denom_c      (wd != 0)                   :69-72
//   p += -lr * exp_avg_c / denom_c              (wd == 0)                   :74
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpf_hip.h"

namespace {

struct AdamArgs {
    float *p; const float *g; float *m, *v, *vmax;
    size_t n;
    float lr, beta1, beta2, omb1, omb2, eps, wd, inv_bc1, inv_bc2;
};

template <bool AMS>
__device__ __forceinline__ void adam_one(const AdamArgs &a, float &p, float g, float &m, float &v, float &vm) {
    m = __fmaf_rn(a.omb1, g, __fmul_rn(m, a.beta1));                         // mul_(beta1).add_(g, alpha = 1 - beta1)
    v = __fmaf_rn(a.omb2, __fmul_rn(g, g), __fmul_rn(v, a.beta2));           // mul_(beta2).addcmul_(g, g, value = 1 - beta2)
    float den;
    if (AMS) { vm = fmaxf(vm, v); den = sqrtf(vm); } else den = sqrtf(v);     // (HIP's __fsqrt_rn is the NATIVE square root; sqrtf under the default flags is correctly rounded)
    den = __fadd_rn(__fmul_rn(den, a.inv_bc2), a.eps);                       // denom / bias_correction2 + eps
    const float q = __fmul_rn(m, a.inv_bc1) / den;                           // (exp_avg / bias_correction1) / denom
    if (a.wd != 0.f) p = __fsub_rn(p, __fmaf_rn(a.lr, q, __fmul_rn(p, a.wd)));   // p -= addcdiv(p * wd, ., ., value = lr)
    else p = __fmaf_rn(-a.lr, q, p);                                         // p.addcdiv_(., ., value = -lr)
}

template <bool AMS>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const size_t n4 = a.n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 p = ((float4 *)a.p)[i], m = ((float4 *)a.m)[i], v = ((float4 *)a.v)[i];
        const float4 g = ((const float4 *)a.g)[i];
        float4 vm = AMS ? ((float4 *)a.vmax)[i] : float4{0.f, 0.f, 0.f, 0.f};
        adam_one<AMS>(a, p.x, g.x, m.x, v.x, vm.x); adam_one<AMS>(a, p.y, g.y, m.y, v.y, vm.y);
        adam_one<AMS>(a, p.z, g.z, m.z, v.z, vm.z); adam_one<AMS>(a, p.w, g.w, m.w, v.w, vm.w);
        ((float4 *)a.p)[i] = p; ((float4 *)a.m)[i] = m; ((float4 *)a.v)[i] = v;
        if (AMS) ((float4 *)a.vmax)[i] = vm;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {                        // tail
        const size_t i = n4 * 4 + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i], vm = AMS ? a.vmax[i] : 0.f;
        adam_one<AMS>(a, p, a.g[i], m, v, vm);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (AMS) a.vmax[i] = vm;
    }
}

}  // namespace

extern "C" int dpf_adam_step(size_t n, float *p, const float *g, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq, double lr,
                             double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                             double bias_correction2, dpf_stream_t stream) {
    if (n == 0) return 0;
    if (!p || !g || !exp_avg || !exp_avg_sq) return DPF_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)max_exp_avg_sq) & 15) return DPF_EINVAL;
    if (!(bias_correction1 != 0.0) || !(bias_correction2 != 0.0)) return DPF_EINVAL;
    // every Python scalar of the reference's ops reaches its kernel as fp32; `1 - beta` and the reciprocals are formed in double first
    AdamArgs a{p, g, exp_avg, exp_avg_sq, max_exp_avg_sq, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1),
               (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(1.0 / bias_correction1), (float)(1.0 / bias_correction2)};
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    if (max_exp_avg_sq) hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
