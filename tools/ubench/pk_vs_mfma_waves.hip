// Micro-test (gfx950), r06 -- the CROSS-WAVE form of the pairing tools/asm_bisect found: in a lone wave, a
// v_pk_fma_f32 directly followed by a v_mfma lost its result about every second launch, and one wait state between the two cured
// it; the vectorised build of csrc/emd.hip with a wait state at each of its seven such pairs still did not repeat at 2048 x 2048,
// where several waves share a SIMD.  Question: does a packed fma of wave A lose its result when wave B -- same SIMD -- issues an
// MFMA right behind it?  No instruction stream of wave A can prevent that.
//   workgroup = 8 waves = 2 per SIMD: waves 0-3 run chains of v_pk_fma_f32 (both op_sel forms of the compiler's text) on random
//   data and compare every chain with v_fma_f32 on the same data; waves 4-7 issue MFMAs -- back to back, or with 0..7 wait states
//   between them so that their issue slots sweep across the packed instructions of the other wave.
//   build: hipcc --offload-arch=gfx950 -O3 pk_vs_mfma_waves.hip -o pk_vs_mfma_waves ; run: ./pk_vs_mfma_waves
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: nobody issues MFMAs (control).  1: MFMAs back to back.  2: an MFMA, then (it & 7) wait states.  3: MFMA waves also run
// packed fmas between their MFMAs (the emd kernels' own mix, every wave alike)
template <int MODE>
__global__ __launch_bounds__(512) void probe(const float *__restrict__ E, int iters, int nset, unsigned long long *bad, float *sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4 && MODE != 3) {
        if (MODE == 0) return;
        h8 a, b;
        for (int u = 0; u < 8; ++u) { a[u] = (_Float16)(0.01f * (lane + u)); b[u] = (_Float16)(0.02f * (lane - u)); }
        f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters * 6; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            if (MODE == 2) { switch (it & 7) { case 1: asm volatile("s_nop 0"); break; case 2: asm volatile("s_nop 1"); break; case 3: asm volatile("s_nop 2"); break;
                             case 4: asm volatile("s_nop 3"); break; case 5: asm volatile("s_nop 4"); break; case 6: asm volatile("s_nop 5"); break; case 7: asm volatile("s_nop 6"); break; default: break; } }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            if (MODE == 2) asm volatile("s_nop 1");
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
        return;
    }
    unsigned long long nb = 0;
    h8 a, b;
    for (int u = 0; u < 8; ++u) { a[u] = (_Float16)(0.01f * (lane + u)); b[u] = (_Float16)(0.02f * (lane - u)); }
    f16v cm = {0};
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 5 + blockIdx.x + wave) % nset;
        const float *e = E + (size_t)set * 20 * 64 + lane;
        float x[16], w[4];
        for (int u = 0; u < 16; ++u) x[u] = e[u * 64];
        for (int u = 0; u < 4; ++u) w[u] = e[(16 + u) * 64];
        f2 acc = {x[0] * 0.5f, x[1] * 0.25f};
        const f2 acc0 = acc;
        // eight packed fmas: terms (x[2k], x[2k+1]) times w[k/2 ...] in the compiler's two forms
        const f2 x01 = {x[0], x[1]}, x23 = {x[2], x[3]}, x45 = {x[4], x[5]}, x67 = {x[6], x[7]}, x89 = {x[8], x[9]}, xab = {x[10], x[11]},
                 xcd = {x[12], x[13]}, xef = {x[14], x[15]}, w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
        if (MODE == 3) cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cm, 0, 0, 0);
        asm volatile("v_pk_fma_f32 %[d], %[s0], %[w01], %[d] op_sel_hi:[1,0,1]\n"
                     "v_pk_fma_f32 %[d], %[s1], %[w01], %[d] op_sel:[0,1,0]\n"
                     "v_pk_fma_f32 %[d], %[s2], %[w23], %[d] op_sel_hi:[1,0,1]\n"
                     "v_pk_fma_f32 %[d], %[s3], %[w23], %[d] op_sel:[0,1,0]\n"
                     : [d] "+v"(acc) : [s0] "v"(x01), [s1] "v"(x23), [s2] "v"(x45), [s3] "v"(x67), [w01] "v"(w01), [w23] "v"(w23));
        if (MODE == 3) cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, cm, 0, 0, 0);
        asm volatile("v_pk_fma_f32 %[d], %[s0], %[w01], %[d] op_sel_hi:[1,0,1]\n"
                     "s_nop 0\n"
                     "v_pk_fma_f32 %[d], %[s1], %[w01], %[d] op_sel:[0,1,0]\n"
                     "s_nop 0\n"
                     "v_pk_fma_f32 %[d], %[s2], %[w23], %[d] op_sel_hi:[1,0,1]\n"
                     "s_nop 0\n"
                     "v_pk_fma_f32 %[d], %[s3], %[w23], %[d] op_sel:[0,1,0]\n"
                     : [d] "+v"(acc) : [s0] "v"(x89), [s1] "v"(xab), [s2] "v"(xcd), [s3] "v"(xef), [w01] "v"(w01), [w23] "v"(w23));
        float r0 = acc0.x, r1 = acc0.y;
        for (int h = 0; h < 2; ++h) {
            const float *xx = x + 8 * h;
            r0 = __builtin_fmaf(xx[0], w[0], r0); r1 = __builtin_fmaf(xx[1], w[0], r1);
            r0 = __builtin_fmaf(xx[2], w[1], r0); r1 = __builtin_fmaf(xx[3], w[1], r1);
            r0 = __builtin_fmaf(xx[4], w[2], r0); r1 = __builtin_fmaf(xx[5], w[2], r1);
            r0 = __builtin_fmaf(xx[6], w[3], r0); r1 = __builtin_fmaf(xx[7], w[3], r1);
        }
        nb += (__float_as_uint(r0) != __float_as_uint(acc.x)) + (__float_as_uint(r1) != __float_as_uint(acc.y));
    }
    if (MODE == 3) sink[blockIdx.x * 512 + threadIdx.x] = cm[0];
    if (nb) atomicAdd(bad, nb);
}

static const char *NAME[] = {"no MFMA anywhere (control)", "the other wave of the SIMD issues MFMAs back to back",
                             "... MFMAs with 0-7 wait states between them", "every wave mixes MFMAs and packed fmas"};

template <int MODE>
void run(const float *E, int nset, unsigned long long *bad, float *sink, int wg_per_cu) {
    hipMemset(bad, 0, 8);
    const int iters = 20000, blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(512), 0, 0, E, iters, nset, bad, sink);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-58s %d workgroup(s) per CU: %.3g packed chains checked, lanes with a wrong result %llu\n", NAME[MODE], wg_per_cu,
           (double)blocks * (MODE == 3 ? 512 : 256) * iters, h);
}

int main() {
    const int nset = 16;
    float *E, *sink; unsigned long long *bad;
    hipMalloc(&E, nset * 20 * 64 * 4); hipMalloc(&sink, 1024 * 512 * 4); hipMalloc(&bad, 8);
    float *h = (float *)malloc(nset * 20 * 64 * 4);
    srand(5);
    for (int i = 0; i < nset * 20 * 64; ++i) h[i] = (rand() % 4 == 0) ? 0.f : ldexpf(0.5f + rand() / (float)RAND_MAX, -(rand() % 28));
    hipMemcpy(E, h, nset * 20 * 64 * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) { run<0>(E, nset, bad, sink, w); run<1>(E, nset, bad, sink, w); run<2>(E, nset, bad, sink, w); run<3>(E, nset, bad, sink, w); }
    return 0;
}
