// Micro-test (gfx950), r06: the instruction sequence of csrc/emd.hip's pair_exponents as the compiler emits it in
// emd_mfma_rows_kernel<4> -- fragments scaled by v_pk_mul_f16 right in front of the MFMAs that read them, two MFMAs chained
// through C -- with the compiler's own wait states (s_nop 1 between the last v_pk_mul_f16 and the second MFMA, nothing between
// the last v_pk_mul_f16 of the first fragment and the first MFMA) and with generous ones, against a reference sequence in which
// every instruction waits for the one before.  The scale is 0.25 (a level below the steepest): a fragment register read before
// its v_pk_mul_f16 has landed holds the UNSCALED value and the result differs.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// registers as in the kernel: zero C v[2:17], A1 v[84:87], A2 v[90:93], B1 v[122:125], B2 v[118:121], D v[66:81]
#define SEQ(N_B1, N_MID, N_B2, N_CHAIN)                                                                                      \
    asm volatile(                                                                                                            \
        "v_mov_b32 v84, %[a10]\n v_mov_b32 v85, %[a11]\n v_mov_b32 v86, %[a12]\n v_mov_b32 v87, %[a13]\n"                     \
        "v_mov_b32 v90, %[a20]\n v_mov_b32 v91, %[a21]\n v_mov_b32 v92, %[a22]\n v_mov_b32 v93, %[a23]\n"                     \
        "v_mov_b32 v122, %[b10]\n v_mov_b32 v123, %[b11]\n v_mov_b32 v124, %[b12]\n v_mov_b32 v125, %[b13]\n"                 \
        "v_mov_b32 v118, %[b20]\n v_mov_b32 v119, %[b21]\n v_mov_b32 v120, %[b22]\n v_mov_b32 v121, %[b23]\n"                 \
        "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"          \
        "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"      \
        "v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n"                                         \
        "s_nop 15\n"                                                                                                         \
        "v_pk_mul_f16 v125, %[s], v125\n v_pk_mul_f16 v124, %[s], v124\n v_pk_mul_f16 v123, %[s], v123\n"                    \
        "v_pk_mul_f16 v122, %[s], v122\n v_pk_mul_f16 v121, %[s], v121\n v_pk_mul_f16 v120, %[s], v120\n" N_B1               \
        "v_mfma_f32_32x32x16_f16 v[66:81], v[84:87], v[122:125], v[2:17]\n" N_MID                                            \
        "v_pk_mul_f16 v119, %[s], v119\n v_pk_mul_f16 v118, %[s], v118\n" N_B2                                               \
        "v_mfma_f32_32x32x16_f16 v[66:81], v[90:93], v[118:121], v[66:81]\n" N_CHAIN                                         \
        "s_nop 15\n s_nop 15\n s_nop 15\n"                                                                                   \
        "v_mov_b32 %[o0], v66\n v_mov_b32 %[o1], v71\n v_mov_b32 %[o2], v76\n v_mov_b32 %[o3], v81\n"                         \
        : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])                                              \
        : [a10] "v"(a1.x), [a11] "v"(a1.y), [a12] "v"(a1.z), [a13] "v"(a1.w), [a20] "v"(a2.x), [a21] "v"(a2.y),               \
          [a22] "v"(a2.z), [a23] "v"(a2.w), [b10] "v"(b1.x), [b11] "v"(b1.y), [b12] "v"(b1.z), [b13] "v"(b1.w),               \
          [b20] "v"(b2.x), [b21] "v"(b2.y), [b22] "v"(b2.z), [b23] "v"(b2.w), [s] "v"(scale)                                  \
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v66",      \
          "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v84",      \
          "v85", "v86", "v87", "v90", "v91", "v92", "v93", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125")

#define W15 "s_nop 15\n s_nop 15\n"

template <int VAR>
__global__ __launch_bounds__(256) void probe(const u4 *__restrict__ A, const u4 *__restrict__ B, int iters, int nset,
                                             unsigned long long *bad) {
    const int lane = threadIdx.x & 63;
    const uint32_t scale = 0x34003400u;          // (0.25, 0.25) fp16
    unsigned long long mism = 0;
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 7 + blockIdx.x + (threadIdx.x >> 6)) % nset, set2 = (set + 3) % nset;
        const u4 a1 = A[set * 64 + lane], a2 = A[set2 * 64 + lane], b1 = B[set * 64 + lane], b2 = B[set2 * 64 + lane];
        float ref[4], o[4];
        { float (&o)[4] = ref; SEQ(W15, W15, W15, W15); }
        if constexpr (VAR == 0) SEQ("", "", "s_nop 1\n", "");                  // the compiler's sequence
        if constexpr (VAR == 1) SEQ("s_nop 7\n", "", "s_nop 1\n", "");         // more in front of the first MFMA only
        if constexpr (VAR == 2) SEQ("", "", "s_nop 7\n", "");                  // more in front of the second MFMA only
        if constexpr (VAR == 3) SEQ("", W15, "s_nop 1\n", "");                 // the first MFMA has finished before B2 is scaled
        if constexpr (VAR == 4) SEQ("s_nop 7\n", "", "s_nop 7\n", "");
        if constexpr (VAR == 5) SEQ("", "", "s_nop 15\n s_nop 15\n", "");
        for (int r = 0; r < 4; ++r) mism += __float_as_uint(ref[r]) != __float_as_uint(o[r]);
    }
    if (mism) atomicAdd(bad, mism);
}

template <int VAR>
void run(const char *what, const u4 *A, const u4 *B, int nset, unsigned long long *bad, int waves_per_simd) {
    (void)hipMemset(bad, 0, 8);
    const int iters = 4000, blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<VAR>), dim3(blocks), dim3(256), 0, 0, A, B, iters, nset, bad);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-78s %d wave(s)/SIMD: %.3g sequences, lanes x registers that differ from the reference: %llu\n", what, waves_per_simd,
           (double)blocks * 4 * iters, h);
}

int main() {
    const int nset = 16;
    u4 *A, *B;
    unsigned long long *bad;
    (void)hipMalloc(&A, nset * 64 * sizeof(u4)); (void)hipMalloc(&B, nset * 64 * sizeof(u4)); (void)hipMalloc(&bad, 8);
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
        (void)hipMemcpy(pass ? (void *)B : (void *)A, h, nset * 64 * sizeof(u4), hipMemcpyHostToDevice);
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("compiler's wait states (0 before MFMA 1, s_nop 1 before MFMA 2)", A, B, nset, bad, w);
        run<1>("s_nop 7 before MFMA 1, s_nop 1 before MFMA 2", A, B, nset, bad, w);
        run<2>("0 before MFMA 1, s_nop 7 before MFMA 2", A, B, nset, bad, w);
        run<3>("MFMA 1 finished before the second fragment is scaled, s_nop 1 before MFMA 2", A, B, nset, bad, w);
        run<4>("s_nop 7 before both", A, B, nset, bad, w);
        run<5>("0 before MFMA 1, 32 wait states before MFMA 2", A, B, nset, bad, w);
    }
    return 0;
}
