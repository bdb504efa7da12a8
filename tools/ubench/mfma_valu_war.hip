// THIS FILE: mfma_war.hip with the overwriting load replaced by four v_mov_b32 into the B registers -- a VALU write right behind
// the MFMAs that read them (the compiler emits exactly that: "v_mfma ... v[118:121] ...; v_lshl_add_u64 v[118:119], ...").
// Micro-test (gfx950), r06 -- the root cause of the run-to-run differences of csrc/emd.hip's matrix-core passes:
// does a VMEM load that OVERWRITES the source registers of MFMAs issued just before it wait for those MFMAs to have read them?
// The compiler assumes so (a source register is dead once its last reader has ISSUED, so it hands v[122:125] -- the scaled B
// fragment four queued MFMAs read -- to the next tile's prefetch: "v_mfma ... v[122:125] ...; global_load_dwordx4 v[122:125]"),
// and the hardware has no interlock for it: MFMAs queue in the matrix pipe (32 cycles each; more with other waves' MFMAs in
// front) and read their A / B operands when they START, a load that hits the L1 / L2 returns in about the same time.
//   per iteration and wave: Q back-to-back independent MFMAs reading B = v[20:23], then a global load INTO v[20:23] of
//   different data (cache-resident), then the results of the FIRST and the LAST of the Q are compared with a reference MFMA
//   executed with nothing behind it.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_war.hip -o mfma_war ; run: ./mfma_war
// Output: per (Q, waves per SIMD, gap): MFMAs checked, results that differ from the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define MF(dst) "v_mfma_f32_32x32x16_f16 " dst ", %[a], v[20:23], 0\n"
#define MOV16(o, base)                                                                                                       \
    "v_mov_b32 %[" o "0], v" #base "\n"

// Q MFMAs (destinations v[40:55] first ... v[136:151] last), then the overwriting load, then everything drains
#define BODY(MFMAS, LASTLO, GAP)                                                                                           \
    asm volatile("global_load_dwordx4 v[20:23], %[pb], off\n s_waitcnt vmcnt(0)\n s_nop 4\n" MFMAS GAP                      \
                 "v_mov_b32 v20, %[x0]\n v_mov_b32 v21, %[x1]\n v_mov_b32 v22, %[x2]\n v_mov_b32 v23, %[x3]\n"                                                              \
                 "s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15\n"                                                    \
                 "v_mov_b32 %[f0], v40\n v_mov_b32 %[f1], v41\n v_mov_b32 %[f2], v47\n v_mov_b32 %[f3], v55\n"              \
                 "v_mov_b32 %[l0], v" #LASTLO "\n v_add_u32 %[l1], 0, v[" #LASTLO "+1]\n"                                    \
                 : [f0] "=&v"(f[0]), [f1] "=&v"(f[1]), [f2] "=&v"(f[2]), [f3] "=&v"(f[3]), [l0] "=&v"(l[0]), [l1] "=&v"(l[1]) \
                 : [a] "v"(a), [pb] "v"(pb), [x0] "v"(x.x), [x1] "v"(x.y), [x2] "v"(x.z), [x3] "v"(x.w)                                                                 \
                 : "memory", "v20", "v21", "v22", "v23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",    \
                   "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62",       \
                   "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",       \
                   "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90",       \
                   "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104",  \
                   "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
                   "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", \
                   "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
                   "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151")

template <int Q, int GAP>
__global__ __launch_bounds__(256) void probe(const u4 *__restrict__ A, const u4 *__restrict__ B, int iters, int nset,
                                             unsigned long long *bad_first, unsigned long long *bad_last) {
    const int lane = threadIdx.x & 63;
    unsigned long long mf = 0, ml = 0;
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 7 + blockIdx.x + (threadIdx.x >> 6)) % nset, set2 = (set + 1) % nset;
        const u4 a = A[set * 64 + lane];
        const u4 *pb = B + set * 64 + lane;
        const u4 x = B[set2 * 64 + lane];
        float f[4], l[2], rf[4], rl[2];
        {   // reference: one MFMA, nothing overwrites its source
            float (&f)[4] = rf; float (&l)[2] = rl;
            BODY(MF("v[40:55]"), 40, "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n");
        }
        if constexpr (Q == 1) { if constexpr (GAP == 0) BODY(MF("v[40:55]"), 40, ""); else BODY(MF("v[40:55]"), 40, "s_nop 7\n"); }
        if constexpr (Q == 2) { if constexpr (GAP == 0) BODY(MF("v[40:55]") MF("v[56:71]"), 56, ""); else BODY(MF("v[40:55]") MF("v[56:71]"), 56, "s_nop 7\n"); }
        if constexpr (Q == 4) {
            if constexpr (GAP == 0) BODY(MF("v[40:55]") MF("v[56:71]") MF("v[72:87]") MF("v[88:103]"), 88, "");
            else BODY(MF("v[40:55]") MF("v[56:71]") MF("v[72:87]") MF("v[88:103]"), 88, "s_nop 7\n");
        }
        if constexpr (Q == 7) {
            if constexpr (GAP == 0) BODY(MF("v[40:55]") MF("v[56:71]") MF("v[72:87]") MF("v[88:103]") MF("v[104:119]") MF("v[120:135]") MF("v[136:151]"), 136, "");
            else BODY(MF("v[40:55]") MF("v[56:71]") MF("v[72:87]") MF("v[88:103]") MF("v[104:119]") MF("v[120:135]") MF("v[136:151]"), 136, "s_nop 7\n");
        }
        for (int r = 0; r < 4; ++r) mf += __float_as_uint(f[r]) != __float_as_uint(rf[r]);
        for (int r = 0; r < 2; ++r) ml += __float_as_uint(l[r]) != __float_as_uint(rl[r]);
    }
    if (mf) atomicAdd(bad_first, mf);
    if (ml) atomicAdd(bad_last, ml);
}

template <int Q, int GAP>
void run(const u4 *A, const u4 *B, int nset, unsigned long long *bad, int waves_per_simd) {
    hipMemset(bad, 0, 16);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;          // 256 CUs x 4 SIMDs: a block of 256 threads = one wave per SIMD
    hipLaunchKernelGGL((probe<Q, GAP>), dim3(blocks), dim3(256), 0, 0, A, B, iters, nset, bad, bad + 1);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("Q = %d MFMAs then four v_mov_b32 onto B%s, %d wave(s)/SIMD: %.3g iterations, lanes with a wrong result: first MFMA %llu, last MFMA %llu\n", Q,
           GAP ? " after s_nop 7" : "", waves_per_simd, (double)blocks * 4 * iters, h[0], h[1]);
}

int main() {
    const int nset = 8;                                // 8 x 1 KiB per operand: resident in every L1
    u4 *A, *B;
    unsigned long long *bad;
    hipMalloc(&A, nset * 64 * sizeof(u4));
    hipMalloc(&B, nset * 64 * sizeof(u4));
    hipMalloc(&bad, 16);
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
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>(A, B, nset, bad, w);
        run<2, 0>(A, B, nset, bad, w);
        run<4, 0>(A, B, nset, bad, w);
        run<4, 1>(A, B, nset, bad, w);
        run<7, 0>(A, B, nset, bad, w);
        run<7, 1>(A, B, nset, bad, w);
    }
    return 0;
}
