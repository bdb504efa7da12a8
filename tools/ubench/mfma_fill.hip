// Micro-benchmark (gfx950): how many plain VALU instructions (and LDS reads) hide in the 32-cycle issue gap of a
// v_mfma_f32_32x32x16_bf16 stream, per filler type, at 1 and 2 waves per SIMD -- the question VERDICT r01 asked
// ("<= 5 plain, independent VALU per MFMA gap; v_pk_fma_f32 vs 2 x v_fma_f32").  Every instruction is a volatile
// inline-asm statement, so the instruction order is exactly the one written here.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill ; run: ./mfma_fill
// Output: per (kind, fillers-per-gap K, waves/SIMD): shader cycles per MFMA (s_memtime, slowest wave of CU 0) and
// wall ns per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// filler kinds
enum { F_FMA = 0, F_SPLIT = 1, F_PKFMA = 2, F_MAX = 3, F_CVT = 4, F_PERM = 5, F_EXP = 6, F_FMAC = 7, F_SUB = 8, F_AND = 9, F_MAXI = 10,
       F_MUL = 11, F_DSR = 12, F_DSV = 13, F_DSV2 = 14, F_PKMAXH = 15, F_MIXLO = 16, F_PKRTZ = 17 };

#define MFMA0 "v_mfma_f32_32x32x16_bf16 %[c0], %[a], %[b], %[c0]\n"
#define MFMA1 "v_mfma_f32_32x32x16_bf16 %[c1], %[a], %[b], %[c1]\n"
#define MFMA2 "v_mfma_f32_32x32x16_bf16 %[c2], %[a], %[b], %[c2]\n"
#define MFMA3 "v_mfma_f32_32x32x16_bf16 %[c3], %[a], %[b], %[c3]\n"

// one filler instruction on chain register i (8 independent chains x0..x7); y = a second operand
#define FILL_FMA(i) "v_fma_f32 %[x" #i "], %[x" #i "], %[k], %[k]\n"
#define FILL_MAX(i) "v_max_f32 %[x" #i "], %[x" #i "], %[k]\n"
#define FILL_CVT(i) "v_cvt_pk_bf16_f32 %[x" #i "], %[x" #i "], %[k]\n"
#define FILL_PERM(i) "v_perm_b32 %[x" #i "], %[x" #i "], %[k], %[sel]\n"
#define FILL_EXP(i) "v_exp_f32 %[x" #i "], %[x" #i "]\n"
#define FILL_PK(i, j) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[kk], %[kk]\n"

__device__ u32x4 g_sink[4];
template <int KIND, int K>
__device__ __forceinline__ void gap(float (&x)[8], double (&p)[4], float k, double kk, uint32_t sel) {
    if (KIND == F_DSR || KIND == F_DSV || KIND == F_DSV2) {
        // ds_read_b128 fillers (conflict-free: lane * 16 B), results never consumed inside the loop; the rest of the gap is
        // the split-mix VALU
        extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
        const uint32_t addr = (uint32_t)(uintptr_t)lds_raw + (threadIdx.x & 63) * 16 + ((threadIdx.x >> 6) & 7) * 1024;
        constexpr int ND = KIND == F_DSR ? K : (KIND == F_DSV ? 1 : 2);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            u32x4 d;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"((i & 7) * 8192));
            asm volatile("" ::"v"(d));
        }
#pragma unroll
        for (int i = 0; i < (KIND == F_DSR ? 0 : K - ND); ++i) {
            float &r = x[i & 7];
            switch (i % 5) {
                case 0: asm volatile("v_max_i32 %0, 0, %0" : "+v"(r)); break;
                case 1: asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r)); break;
                case 2: asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(k)); break;
                case 3: asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(sel)); break;
                default: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(k)); break;
            }
        }
        return;
    }
    // K filler instructions (for F_PKFMA: K/2 packed instructions = the work of K scalar ones)
    if (KIND == F_PKFMA) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i) {
            switch (i & 3) {
                case 0: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[0]) : "v"(kk)); break;
                case 1: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[1]) : "v"(kk)); break;
                case 2: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[2]) : "v"(kk)); break;
                default: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[3]) : "v"(kk)); break;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float &r = x[i & 7];
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(k));
        if (KIND == F_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(k));
        if (KIND == F_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(r) : "v"(k));
        if (KIND == F_SUB) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(r) : "v"(k));
        if (KIND == F_AND) asm volatile("v_and_b32_e32 %0, 0xffff0000, %0" : "+v"(r));
        if (KIND == F_MAXI) asm volatile("v_max_i32_e32 %0, 0, %0" : "+v"(r));
        if (KIND == F_MUL) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r) : "v"(k));
        if (KIND == F_PKMAXH) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r) : "v"(k));
        if (KIND == F_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %1 op_sel_hi:[1,0,0] clamp" : "+v"(r) : "v"(k));
        if (KIND == F_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r) : "v"(k));
        if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(k));
        if (KIND == F_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(sel));
        if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
        if (KIND == F_SPLIT) {   // the relu / hi-lo split mix of flow_kernel: max_i32, and, sub, perm, cvt_pk in rotation
            switch (i % 5) {
                case 0: asm volatile("v_max_i32 %0, 0, %0" : "+v"(r)); break;
                case 1: asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r)); break;
                case 2: asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(k)); break;
                case 3: asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(sel)); break;
                default: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(k)); break;
            }
        }
    }
}

// MODE 0: every wave runs [MFMA, K fillers] x 64 per iteration on NACC accumulators (round robin).
// MODE 1: wave-specialised -- waves of the first half of the workgroup run MFMA only, the second half run the
//         fillers only (same totals per SIMD as MODE 0 at 2 waves/SIMD), to see whether two waves' streams overlap.
// MODE 2: MFMA phase then filler phase in program order (not interleaved), every wave; with 2 waves/SIMD the
//         second half of the workgroup runs the phases in the opposite order (phase-shifted partners).
template <int KIND, int K, int NACC, int MODE>
__global__ __launch_bounds__(1024) void kern(float *out, int iters, float seed, unsigned long long *ticks) {
    float x[8];
    double p[4];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) p[i] = (double)seed + i;
    const float k = seed * 0.999f;
    const double kk = (double)seed * 1.0001;
    const uint32_t sel = 0x07060302u;
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) c[j][i] = seed * i;
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool second = wave >= nw / 2 && nw >= 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a), "v"(b));
                gap<KIND, K>(x, p, k, kk, sel);
            }
        } else if (MODE == 3) {   // no MFMA at all: the fillers' own issue rate
#pragma unroll
            for (int m = 0; m < 64; ++m) gap<KIND, K>(x, p, k, kk, sel);
        } else if (MODE == 1) {
            if (!second) {
#pragma unroll
                for (int m = 0; m < 64; ++m)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a), "v"(b));
            } else {
#pragma unroll
                for (int m = 0; m < 64; ++m) gap<KIND, K>(x, p, k, kk, sel);
            }
        } else {
            auto mph = [&]() {
#pragma unroll
                for (int m = 0; m < 32; ++m)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a), "v"(b));
            };
            auto vph = [&]() {
#pragma unroll
                for (int m = 0; m < 32; ++m) gap<KIND, K>(x, p, k, kk, sel);
            };
            if (!second) { mph(); vph(); mph(); vph(); }
            else { vph(); mph(); vph(); mph(); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += (float)p[i];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

static float *d_out;
static unsigned long long *d_ticks;

template <int KIND, int K, int NACC, int MODE>
void run(const char *name, int wps) {
    const int iters = 200, threads = wps * 256, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {   // first run warms up
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<KIND, K, NACC, MODE>), dim3(blocks), dim3(threads), 65536, 0, d_out, iters, 1.0f, d_ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[16] = {0};
    hipMemcpy(t, d_ticks, sizeof(t), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = t[w] > mx ? t[w] : mx;
    // MFMAs per SIMD: MODE 0/2: wps waves x 64 per iter; MODE 1: half the waves
    const double mf = (MODE == 1 ? wps / 2.0 : (double)wps) * 64.0 * iters;
    printf("%-7s K=%2d acc=%d mode=%d waves/SIMD=%d : %7.1f cyc/MFMA/SIMD (s_memtime, slowest wave)  %6.2f ns/MFMA/SIMD wall  "
           "[fillers per MFMA per SIMD-stream: %d]\n",
           name, K, NACC, MODE, wps, (double)mx / mf * 1.0, ms * 1e6 / mf, K);
}

#define SWEEP(KIND, NAME, NACC, MODE, WPS)       \
    run<KIND, 0, NACC, MODE>(NAME, WPS);         \
    run<KIND, 2, NACC, MODE>(NAME, WPS);         \
    run<KIND, 4, NACC, MODE>(NAME, WPS);         \
    run<KIND, 5, NACC, MODE>(NAME, WPS);         \
    run<KIND, 6, NACC, MODE>(NAME, WPS);         \
    run<KIND, 8, NACC, MODE>(NAME, WPS);         \
    run<KIND, 10, NACC, MODE>(NAME, WPS);        \
    run<KIND, 12, NACC, MODE>(NAME, WPS);        \
    run<KIND, 16, NACC, MODE>(NAME, WPS);

#define PURE(KIND, NAME)                                            \
    for (int wps = 1; wps <= 4; wps *= 2) run<KIND, 16, 2, 3>(NAME, wps);

int main(int argc, char **argv) {
    hipMalloc(&d_out, 256 * 1024 * sizeof(float));
    hipMalloc(&d_ticks, 64 * sizeof(unsigned long long));
    if (argc > 1 && argv[1][0] == 'h') {   // the fp16 split's instructions as gap fillers
        SWEEP(F_PKMAXH, "pkmaxh", 2, 0, 2)
        SWEEP(F_MIXLO, "mixlo", 2, 0, 2)
        SWEEP(F_PKRTZ, "pkrtz", 2, 0, 2)
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {   // LDS reads as gap fillers
        for (int wps = 1; wps <= 2; ++wps) {
            SWEEP(F_DSR, "dsr", 2, 0, wps)      // K ds_read_b128 per MFMA gap, nothing else
            SWEEP(F_DSV, "ds1+v", 2, 0, wps)    // 1 ds_read_b128 + (K-1) VALU per gap
            SWEEP(F_DSV2, "ds2+v", 2, 0, wps)   // 2 ds_read_b128 + (K-2) VALU per gap
        }
        return 0;
    }
    if (argc > 1) {   // pure-VALU issue rates (mode 3: cyc column = cycles per 16 instructions per SIMD-"slot")
        PURE(F_FMA, "fma") PURE(F_FMAC, "fmac") PURE(F_SUB, "sub") PURE(F_MUL, "mul") PURE(F_AND, "and") PURE(F_MAXI, "maxi")
        PURE(F_MAX, "maxf") PURE(F_PERM, "perm") PURE(F_CVT, "cvtpk") PURE(F_PKFMA, "pkfma") PURE(F_EXP, "exp") PURE(F_SPLIT, "split")
        SWEEP(F_FMAC, "fmac", 2, 0, 1)
        SWEEP(F_FMAC, "fmac", 2, 0, 2)
        SWEEP(F_SUB, "sub", 2, 0, 2)
        SWEEP(F_AND, "and", 2, 0, 2)
        SWEEP(F_MAXI, "maxi", 2, 0, 2)
        SWEEP(F_PERM, "perm", 2, 0, 2)
        return 0;
    }
    printf("# s_memtime ticks are 100 MHz-class constant-rate on some parts: compare the K sweep RELATIVE to K=0, and wall ns.\n");
    for (int wps = 1; wps <= 2; ++wps) {
        SWEEP(F_FMA, "fma", 2, 0, wps)
        SWEEP(F_SPLIT, "split", 2, 0, wps)
        SWEEP(F_PKFMA, "pkfma", 2, 0, wps)
        SWEEP(F_CVT, "cvtpk", 2, 0, wps)
        SWEEP(F_EXP, "exp", 2, 0, wps)
    }
    SWEEP(F_FMA, "fma", 4, 0, 1)
    SWEEP(F_FMA, "fma", 4, 0, 2)
    // wave-specialised and phase-shifted partners (2 waves/SIMD only)
    SWEEP(F_FMA, "fma", 2, 1, 2)
    SWEEP(F_FMA, "fma", 2, 2, 2)
    SWEEP(F_SPLIT, "split", 2, 2, 2)
    SWEEP(F_FMA, "fma", 2, 2, 1)
    return 0;
}
