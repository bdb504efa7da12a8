// Micro-test (gfx950): is a v_mfma_f32_32x32x16_f16 whose DESTINATION registers overlap those of source A or source B safe?
// The compiler emits such instructions for the first MFMA of a chain (C = 0, a source dying at the instruction):
//     v_mfma_f32_32x32x16_f16 v[2:17], v[70:73], v[2:5], 0
// and csrc/emd.hip's sparse-regime pass 2 returned sums that differed from run to run until its MFMAs got an early-clobber
// destination.  Here every variant is a fixed-register inline-asm sequence around ONE MFMA -- the sources are produced by
// v_pk_mul_f16 right in front of it, as in that kernel, and the result is read by v_exp_f32 behind the compiler's own number of
// wait states (s_nop 10) or behind a generous 19 -- run `iters` times per wave on the same operands and compared bit for bit
// with a reference MFMA whose destination is kept apart.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap ; run: ./mfma_overlap
// Output: per (variant, waves per SIMD, wait states): MFMAs executed, results that differ from the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// variants: where the destination v[2:17] meets a source
enum { V_APART = 0, V_B_HEAD = 1, V_B_MID = 2, V_A_HEAD = 3, V_A_MID = 4, V_COUNT = 5 };

#define SEQ(AREG, BREG, NOPS)                                                                                          \
    "v_pk_mul_f16 " AREG "0, %[one], %[a0]\n v_pk_mul_f16 " AREG "1, %[one], %[a1]\n"                                    \
    "v_pk_mul_f16 " AREG "2, %[one], %[a2]\n v_pk_mul_f16 " AREG "3, %[one], %[a3]\n"                                    \
    "v_pk_mul_f16 " BREG "0, %[one], %[b0]\n v_pk_mul_f16 " BREG "1, %[one], %[b1]\n"                                    \
    "v_pk_mul_f16 " BREG "2, %[one], %[b2]\n v_pk_mul_f16 " BREG "3, %[one], %[b3]\n"

// the registers are fixed: destination v[2:17]; sources at v[2:5] / v[8:11] (overlapping) or v[40:43] / v[44:47] (apart)
#define RUN(ASRC, BSRC, A0, A1, A2, A3, B0, B1, B2, B3, NOPS)                                                            \
    asm volatile("v_pk_mul_f16 " A0 ", %[one], %[a0]\n v_pk_mul_f16 " A1 ", %[one], %[a1]\n"                             \
                 "v_pk_mul_f16 " A2 ", %[one], %[a2]\n v_pk_mul_f16 " A3 ", %[one], %[a3]\n"                             \
                 "v_pk_mul_f16 " B0 ", %[one], %[b0]\n v_pk_mul_f16 " B1 ", %[one], %[b1]\n"                             \
                 "v_pk_mul_f16 " B2 ", %[one], %[b2]\n v_pk_mul_f16 " B3 ", %[one], %[b3]\n"                             \
                 "s_nop 1\n"                                                                                             \
                 "v_mfma_f32_32x32x16_f16 v[2:17], " ASRC ", " BSRC ", 0\n" NOPS                                         \
                 "v_exp_f32 %[o0], v2\n v_exp_f32 %[o1], v3\n v_exp_f32 %[o2], v4\n v_exp_f32 %[o3], v5\n"               \
                 "v_exp_f32 %[o4], v6\n v_exp_f32 %[o5], v7\n v_exp_f32 %[o6], v8\n v_exp_f32 %[o7], v9\n"               \
                 "v_exp_f32 %[o8], v10\n v_exp_f32 %[o9], v11\n v_exp_f32 %[o10], v12\n v_exp_f32 %[o11], v13\n"         \
                 "v_exp_f32 %[o12], v14\n v_exp_f32 %[o13], v15\n v_exp_f32 %[o14], v16\n v_exp_f32 %[o15], v17\n"       \
                 : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]),             \
                   [o5] "=&v"(o[5]), [o6] "=&v"(o[6]), [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]),             \
                   [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), [o12] "=&v"(o[12]), [o13] "=&v"(o[13]), [o14] "=&v"(o[14]),   \
                   [o15] "=&v"(o[15])                                                                                    \
                 : [one] "v"(one), [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w), [b0] "v"(b.x),           \
                   [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w)                                                           \
                 : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16",      \
                   "v17", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")

#define NOP_COMPILER "s_nop 10\n"
#define NOP_LONG "s_nop 15\n s_nop 2\n"

template <int VARIANT, bool LONG>
__device__ __forceinline__ void one(u4 a, u4 b, uint32_t one, float (&o)[16]) {
    if constexpr (VARIANT == V_APART) {
        if constexpr (LONG) RUN("v[40:43]", "v[44:47]", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", NOP_LONG);
        else RUN("v[40:43]", "v[44:47]", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", NOP_COMPILER);
    } else if constexpr (VARIANT == V_B_HEAD) {
        if constexpr (LONG) RUN("v[40:43]", "v[2:5]", "v40", "v41", "v42", "v43", "v2", "v3", "v4", "v5", NOP_LONG);
        else RUN("v[40:43]", "v[2:5]", "v40", "v41", "v42", "v43", "v2", "v3", "v4", "v5", NOP_COMPILER);
    } else if constexpr (VARIANT == V_B_MID) {
        if constexpr (LONG) RUN("v[40:43]", "v[8:11]", "v40", "v41", "v42", "v43", "v8", "v9", "v10", "v11", NOP_LONG);
        else RUN("v[40:43]", "v[8:11]", "v40", "v41", "v42", "v43", "v8", "v9", "v10", "v11", NOP_COMPILER);
    } else if constexpr (VARIANT == V_A_HEAD) {
        if constexpr (LONG) RUN("v[2:5]", "v[44:47]", "v2", "v3", "v4", "v5", "v44", "v45", "v46", "v47", NOP_LONG);
        else RUN("v[2:5]", "v[44:47]", "v2", "v3", "v4", "v5", "v44", "v45", "v46", "v47", NOP_COMPILER);
    } else {
        if constexpr (LONG) RUN("v[8:11]", "v[44:47]", "v8", "v9", "v10", "v11", "v44", "v45", "v46", "v47", NOP_LONG);
        else RUN("v[8:11]", "v[44:47]", "v8", "v9", "v10", "v11", "v44", "v45", "v46", "v47", NOP_COMPILER);
    }
}

template <int VARIANT, bool LONG>
__global__ __launch_bounds__(256) void probe(const u4 *__restrict__ A, const u4 *__restrict__ B, int iters, int nset,
                                             unsigned long long *bad) {
    const int lane = threadIdx.x & 63;
    const uint32_t one_h = 0x3c003c00u;      // (1.0, 1.0) fp16
    unsigned long long mism = 0;
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 7 + blockIdx.x + (threadIdx.x >> 6)) % nset;
        const u4 a = A[set * 64 + lane], b = B[set * 64 + lane];
        float ref[16], o[16];
        one<V_APART, true>(a, b, one_h, ref);
        one<VARIANT, LONG>(a, b, one_h, o);
#pragma unroll
        for (int r = 0; r < 16; ++r) mism += __float_as_uint(ref[r]) != __float_as_uint(o[r]);
    }
    if (mism) atomicAdd(bad, mism);
}

template <int VARIANT, bool LONG>
void run(const char *name, const u4 *A, const u4 *B, int nset, unsigned long long *bad, int waves_per_simd) {
    hipMemset(bad, 0, 8);
    const int iters = 20000;
    // 256 CUs x 4 SIMDs: blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<VARIANT, LONG>), dim3(blocks), dim3(256), 0, 0, A, B, iters, nset, bad);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-34s %d wave(s)/SIMD, %s: %.3g MFMAs (x 1024 results), results differing from the reference: %llu\n", name, waves_per_simd,
           LONG ? "19 wait states" : "11 wait states", (double)blocks * 4 * iters, h);
}

int main() {
    const int nset = 64;
    u4 *A, *B;
    unsigned long long *bad;
    hipMalloc(&A, nset * 64 * sizeof(u4));
    hipMalloc(&B, nset * 64 * sizeof(u4));
    hipMalloc(&bad, 8);
    // fp16 operands of moderate size (|x| < 2): random bit patterns with the exponent field forced into [13, 15]
    uint32_t *h = (uint32_t *)malloc(nset * 64 * sizeof(u4));
    for (int pass = 0; pass < 2; ++pass) {
        srand(7 + pass);
        for (int i = 0; i < nset * 64 * 4; ++i) {
            uint32_t w = 0;
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t mant = rand() & 0x3ff, ex = 13 + rand() % 3, sg = rand() & 1;
                w |= ((sg << 15) | (ex << 10) | mant) << (16 * hh);
            }
            h[i] = w;
        }
        hipMemcpy(pass ? (void *)B : (void *)A, h, nset * 64 * sizeof(u4), hipMemcpyHostToDevice);
    }
    for (int w = 1; w <= 2; ++w) {
        run<V_APART, false>("destination apart", A, B, nset, bad, w);
        run<V_B_HEAD, false>("source B = v[2:5]  (head of dst)", A, B, nset, bad, w);
        run<V_B_HEAD, true>("source B = v[2:5]  (head of dst)", A, B, nset, bad, w);
        run<V_B_MID, false>("source B = v[8:11] (inside dst)", A, B, nset, bad, w);
        run<V_B_MID, true>("source B = v[8:11] (inside dst)", A, B, nset, bad, w);
        run<V_A_HEAD, false>("source A = v[2:5]  (head of dst)", A, B, nset, bad, w);
        run<V_A_HEAD, true>("source A = v[2:5]  (head of dst)", A, B, nset, bad, w);
        run<V_A_MID, false>("source A = v[8:11] (inside dst)", A, B, nset, bad, w);
        run<V_A_MID, true>("source A = v[8:11] (inside dst)", A, B, nset, bad, w);
    }
    return 0;
}
