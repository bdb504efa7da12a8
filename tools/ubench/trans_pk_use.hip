// Micro-test (gfx950), r06 -- THE root cause of the run-to-run differences of csrc/emd.hip's matrix-core passes: a
// PACKED fp32 VALU instruction that consumes the results of transcendental instructions issued just before it.  The SLP vectoriser
// turns the passes' independent accumulations  s[r] = fma(exp2(e[r]), w, s[r])  into
//     v_exp_f32 v68, v68 ; v_exp_f32 v69, v69 ; v_exp_f32 v34, v34 ; ... ; v_pk_fma_f32 v[200:201], v[68:69], v[206:207], v[200:201]
// (emd_mfma_rows_kernel<4>, -O3 without -fno-slp-vectorize).  The compiler's hazard recogniser keeps ONE wait state between a
// transcendental result and its VALU use; this test measures, for T back-to-back v_exp_f32 followed by a consumer of the first
// two results with G instructions of gap, how often the consumer sees a stale register -- for the packed consumer and for two
// plain v_fma_f32.
//   build: hipcc --offload-arch=gfx950 -O3 trans_pk_use.hip -o trans_pk_use ; run: ./trans_pk_use
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// v[60..67]: exp results (preset to a poison value); consumer reads v[60:61]
#define EXP(n) "v_exp_f32 v" #n ", %[x" #n "]\n"
#define PRE "v_mov_b32 v60, %[p]\n v_mov_b32 v61, %[p]\n v_mov_b32 v62, %[p]\n v_mov_b32 v63, %[p]\n v_mov_b32 v64, %[p]\n v_mov_b32 v65, %[p]\n v_mov_b32 v66, %[p]\n v_mov_b32 v67, %[p]\n s_nop 7\n"
#define PK "v_pk_fma_f32 %[o], v[60:61], %[w], %[z]\n"
#define SC "v_fma_f32 %[o0], v60, %[w0], %[z0]\n v_fma_f32 %[o1], v61, %[w0], %[z0]\n"
#define IO_PK : [o] "=&v"(o) : [p] "v"(poison), [w] "v"(w), [z] "v"(z), [x60] "v"(x[0]), [x61] "v"(x[1]), [x62] "v"(x[2]), [x63] "v"(x[3]), [x64] "v"(x[4]), [x65] "v"(x[5]), [x66] "v"(x[6]), [x67] "v"(x[7]) : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67"
#define IO_SC : [o0] "=&v"(o.x), [o1] "=&v"(o.y) : [p] "v"(poison), [w0] "v"(w.x), [z0] "v"(z.x), [x60] "v"(x[0]), [x61] "v"(x[1]), [x62] "v"(x[2]), [x63] "v"(x[3]), [x64] "v"(x[4]), [x65] "v"(x[5]), [x66] "v"(x[6]), [x67] "v"(x[7]) : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67"

typedef float f2 __attribute__((ext_vector_type(2)));
#define MF2 "v_mfma_f32_32x32x16_f16 v[100:115], v[70:73], v[74:77], 0\n v_mfma_f32_32x32x16_f16 v[116:131], v[70:73], v[74:77], 0\n"
#define MCLOB , "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131"
#define IO_PKM IO_PK MCLOB
#define IO_SCM IO_SC MCLOB

#define T2 EXP(60) EXP(61)
#define T5 EXP(60) EXP(61) EXP(62) EXP(63) EXP(64)
#define T8 EXP(60) EXP(61) EXP(62) EXP(63) EXP(64) EXP(65) EXP(66) EXP(67)
// the consumer's operands are first (v60) and second (v61) result; with T = 5 / 8 the later exps sit between def and use

template <int VAR>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ X, int iters, unsigned long long *bad) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mism = 0;
    const float poison = 12345.f;
    const f2 w = {0.75f, 0.75f}, z = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = X[(tid * 8 + i + it * 13) & 4095];
        f2 o, r;
        { f2 &o = r; asm volatile(PRE T8 "s_nop 15\n s_nop 15\n s_nop 15\n" SC IO_SC); }          // reference
        if constexpr (VAR == 0) asm volatile(PRE T2 "s_nop 0\n" PK IO_PK);                          // the compiler's single wait state
        if constexpr (VAR == 1) asm volatile(PRE T2 "s_nop 0\n" SC IO_SC);
        if constexpr (VAR == 2) asm volatile(PRE T5 PK IO_PK);                                      // the kernel's sequence: 3 exps in between
        if constexpr (VAR == 3) asm volatile(PRE T5 SC IO_SC);
        if constexpr (VAR == 4) asm volatile(PRE T8 PK IO_PK);
        if constexpr (VAR == 5) asm volatile(PRE T8 SC IO_SC);
        if constexpr (VAR == 6) asm volatile(PRE T5 "s_nop 3\n" PK IO_PK);
        if constexpr (VAR == 7) asm volatile(PRE T8 "s_nop 7\n" PK IO_PK);
        // ... the same with MFMAs in flight (the packed fp32 instructions share hardware with the matrix pipe: +16 cycles each beside
        // an MFMA, tools/ubench/mfma_fill.hip): two independent MFMAs issued right in front of the exps / of the consumer
        if constexpr (VAR == 8) asm volatile(PRE MF2 T5 PK IO_PKM);
        if constexpr (VAR == 9) asm volatile(PRE MF2 T5 SC IO_SCM);
        if constexpr (VAR == 10) asm volatile(PRE T5 MF2 PK IO_PKM);
        if constexpr (VAR == 11) asm volatile(PRE T5 MF2 SC IO_SCM);
        if constexpr (VAR == 12) asm volatile(PRE MF2 T8 MF2 PK IO_PKM);
        if constexpr (VAR == 13) asm volatile(PRE MF2 T8 MF2 SC IO_SCM);
        mism += (__float_as_uint(o.x) != __float_as_uint(r.x)) + (__float_as_uint(o.y) != __float_as_uint(r.y));
    }
    if (mism) atomicAdd(bad, mism);
}

template <int VAR>
void run(const char *what, const float *X, unsigned long long *bad, int waves_per_simd) {
    (void)hipMemset(bad, 0, 8);
    const int iters = 2000, blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<VAR>), dim3(blocks), dim3(256), 0, 0, X, iters, bad);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-64s %d wave(s)/SIMD: %.3g uses, stale: %llu\n", what, waves_per_simd, 2.0 * blocks * 256 * iters, h);
}

int main() {
    float *X, h[4096];
    unsigned long long *bad;
    (void)hipMalloc(&X, sizeof(h)); (void)hipMalloc(&bad, 8);
    srand(3);
    for (int i = 0; i < 4096; ++i) h[i] = -8.f * (float)rand() / (float)RAND_MAX;
    (void)hipMemcpy(X, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("2 exp, s_nop 0, v_pk_fma_f32 of both", X, bad, w);
        run<1>("2 exp, s_nop 0, two v_fma_f32", X, bad, w);
        run<2>("5 exp, v_pk_fma_f32 of the first two (the kernel's sequence)", X, bad, w);
        run<3>("5 exp, two v_fma_f32 of the first two", X, bad, w);
        run<4>("8 exp, v_pk_fma_f32 of the first two", X, bad, w);
        run<5>("8 exp, two v_fma_f32 of the first two", X, bad, w);
        run<6>("5 exp, s_nop 3, v_pk_fma_f32", X, bad, w);
        run<7>("8 exp, s_nop 7, v_pk_fma_f32", X, bad, w);
        run<8>("2 MFMAs, 5 exp, v_pk_fma_f32", X, bad, w);
        run<9>("2 MFMAs, 5 exp, two v_fma_f32", X, bad, w);
        run<10>("5 exp, 2 MFMAs, v_pk_fma_f32", X, bad, w);
        run<11>("5 exp, 2 MFMAs, two v_fma_f32", X, bad, w);
        run<12>("2 MFMAs, 8 exp, 2 MFMAs, v_pk_fma_f32", X, bad, w);
        run<13>("2 MFMAs, 8 exp, 2 MFMAs, two v_fma_f32", X, bad, w);
    }
    return 0;
}
