// Micro-test (gfx950): does a VALU instruction that reads the result of v_exp_f32 get the right value when several waves of a
// SIMD keep the transcendental unit busy?  (r05: the matrix-core approx-EMD passes -- 16 v_exp_f32 per MFMA, their results read by
// v_pk_fma_f32 a few instructions later -- returned different bits from run to run in some builds, only with two or more waves per
// SIMD, more often the longer the kernel ran, and much more often when the v_exp_f32 were volatile asm statements followed by
// their readers at the compiler-unknown minimum distance.)
// Every wave runs, ITERS times: 16 x (v_exp_f32 ; <gap> ; v_fma_f32 reading it), gap = 1 wait state (what the compiler inserts),
// and the same with a gap of 32 wait states as the reference; sums compared bit for bit.
//   build: hipcc --offload-arch=gfx950 -O3 trans_contention.hip -o trans_contention ; run: ./trans_contention
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define STEP(GAP)                                                                                  \
    asm volatile("v_exp_f32 %0, %2\n" GAP "v_fma_f32 %1, %0, %3, %1" : "=&v"(e), "+v"(s) : "v"(x), "v"(w));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float *in, int iters, unsigned long long *bad, float *out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = in[tid & 4095], w = 0.37f + 0.001f * (tid & 63);
    unsigned long long mism = 0;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        float sref = 0.f, s = 0.f, e, x;
#pragma unroll
        for (int r = 0; r < 16; ++r) { x = x0 - 0.37f * r - 0.01f * (it & 15); float &S = sref; (void)S; asm volatile("v_exp_f32 %0, %2\ns_nop 15\ns_nop 15\nv_fma_f32 %1, %0, %3, %1" : "=&v"(e), "+v"(sref) : "v"(x), "v"(w)); }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x = x0 - 0.37f * r - 0.01f * (it & 15);
            if (MODE == 0) { STEP("s_nop 0\n") }            // one wait state between the exp and its reader
            else if (MODE == 1) { STEP("s_nop 3\n") }
            else { s = __builtin_fmaf(__builtin_amdgcn_exp2f(x), w, s); }      // compiler's own scheduling and wait states
        }
        mism += __float_as_uint(s) != __float_as_uint(sref);
        keep += s;
    }
    if (mism) atomicAdd(bad, mism);
    if (keep == 123.456f) out[0] = keep;
}

template <int MODE>
void run(const char *name, const float *in, unsigned long long *bad, float *out, int waves_per_simd) {
    hipMemset(bad, 0, 8);
    const int iters = 20000;
    // one block of 1024 threads = 16 waves = 4 per SIMD of a CU; fewer waves per SIMD: smaller blocks
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * (256 * waves_per_simd / threads);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, in, iters, bad, out);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s)/SIMD: %.3g (exp, reader) groups of 16, sums differing from the reference: %llu\n", name, waves_per_simd,
           (double)blocks * (threads / 64) * iters, h);
}

int main() {
    float *in, *out;
    unsigned long long *bad;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4); hipMalloc(&bad, 8);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = -0.003f * i;          // arguments from 0 down to -12 (and lower inside the kernel)
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 8}) {
        run<0>("asm: v_exp_f32, 1 wait state, v_fma_f32", in, bad, out, w);
        run<1>("asm: v_exp_f32, 4 wait states, v_fma_f32", in, bad, out, w);
        run<2>("compiler: exp2f builtin + fmaf", in, bad, out, w);
    }
    return 0;
}
