// Chamfer nearest-neighbour distance for gfx950 (MI355X): forward (min + first
// argmin, bit-exact contract) and backward (scatter through the argmin).
//
// Replaces NmDistanceKernel / NmDistanceGradKernel of the reference
// (lib/metrics/pytorch_structural_losses/src/nndistance.cu:2-154).
//
// Forward design (VALU-bound brute force, 2*B*n*m pair evaluations):
//  * one wave owns 128 contiguous query points (two per lane, held in
//    registers as packed f32 pairs so the distance runs on v_pk_{add,mul}_f32);
//  * candidate points are wave-uniform, so they are fetched with SCALAR loads
//    straight into SGPRs (no LDS, no bank conflicts, no vector-memory issue
//    slots) and broadcast into the packed ops;
//  * the inner loop only tracks the running MINIMUM per chunk of 8 candidates
//    (v_min), and remembers which chunk first achieved the best value; the
//    winning chunk is re-scanned once at the end to recover the first index.
//    That is arithmetically identical to the reference's strict '<' scan in
//    ascending k (nndistance.cu:26,116) but costs ~5 instead of ~8 VALU ops
//    per pair;
//  * small problems split the candidate range over the waves of a workgroup
//    (KS) so the launch still fills 256 CUs x 4 SIMDs; partial results merge
//    in LDS in ascending slice order with the same strict '<'.
//  d = (dx*dx + dy*dy) + dz*dz, dx = candidate - query, every operation rounded
//  separately: this file is compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dpf_hip.h"
#include "lds_attr.h"
#include "zero_fill.h"
#include "nn_refscan.h"

#pragma clang fp contract(off)

// -DDPF_NN_ABLATE=<mask>: timing experiments of tools/nn_ablate.py (results are garbage): 1 no candidate exponent check,
// 2 no first-index rescan, 4 no scan loop, 8 no query loads
#ifndef DPF_NN_ABLATE
#define DPF_NN_ABLATE 0
#endif

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at dword alignment (one global_load_dwordx4)

constexpr int CH = 8;            // candidates per min-chunk
constexpr int NWAVES = 4;        // waves per workgroup (default; the small-problem variants use 8 and 16)
constexpr int QPW = 128;         // query points per wave (2 per lane)

struct NNDir {
    const float *q;   // (B, nq, 3) queries
    const float *c;   // (B, nc, 3) candidates
    float *dist;      // (B, nq)
    int *idx;         // (B, nq)
    int nq, nc;
    long qstride, cstride;   // floats between consecutive clouds (0 = one cloud broadcast over the batch)
};

struct NNArgs {
    NNDir d[2];
    // nn_small_kernel only (dpf_nn_small_cd): per-workgroup sums of the distances, one ticket per cloud, cd (B,) -- the
    // per-cloud Chamfer reduction finished inside the search launch, as nnm_kernel does it (chamfer_mfma.hip)
    float *part = nullptr;
    unsigned *ticket = nullptr;
    float *cd = nullptr;
};

__device__ __forceinline__ f2 pair_dist(float sx, float sy, float sz, f2 qx, f2 qy, f2 qz) {
    const f2 dx = sx - qx, dy = sy - qy, dz = sz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}

// Candidate coordinates live in SGPR *pairs* (64-bit scalar loads); element e of
// a chunk is the (e&1) half of pair e>>1.  v_pk_add_f32 broadcasts that half to
// both packed lanes through op_sel, so no SGPR copies or VGPR splats are needed
// (hipcc otherwise emits an s_mov per odd element, which also drags the
// scalar-load wait to the top of the chunk).
template <int HALF>
__device__ __forceinline__ f2 bsub(unsigned long long pr, f2 q) {   // {s,s} - q
    f2 r;
    if constexpr (HALF == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(pr), "v"(q));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(pr), "v"(q));
    return r;
}

template <int U>   // candidate U of a chunk held as 12 pairs
__device__ __forceinline__ f2 pair_dist_sp(const unsigned long long (&pr)[CH * 3 / 2], f2 qx, f2 qy, f2 qz) {
    const f2 dx = bsub<(3 * U + 0) & 1>(pr[(3 * U + 0) >> 1], qx);
    const f2 dy = bsub<(3 * U + 1) & 1>(pr[(3 * U + 1) >> 1], qy);
    const f2 dz = bsub<(3 * U + 2) & 1>(pr[(3 * U + 2) >> 1], qz);
    return (dx * dx + dy * dy) + dz * dz;
}

template <int U>
__device__ __forceinline__ void chunk_min_sp(const unsigned long long (&pr)[CH * 3 / 2], f2 qx, f2 qy, f2 qz, f2 &dm) {
    if constexpr (U < CH) {
        const f2 d = pair_dist_sp<U>(pr, qx, qy, qz);
        dm.x = fminf(dm.x, d.x);
        dm.y = fminf(dm.y, d.y);
        chunk_min_sp<U + 1>(pr, qx, qy, qz, dm);
    }
}

__device__ __forceinline__ float one_dist(float cx, float cy, float cz, float qx, float qy, float qz) {
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}

template <int KS, int NWAVES = 4>
__global__ __launch_bounds__(NWAVES * 64) void nn_kernel(NNArgs args) {
    constexpr int QG = NWAVES / KS;          // query groups per workgroup
    const NNDir A = args.d[blockIdx.z];
    const int nq = A.nq, nc = A.nc;
    if ((int)blockIdx.x * QG * QPW >= nq) return;   // this direction has fewer query tiles
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = wave / KS, ks = wave % KS;
    const float *__restrict__ q = A.q + (size_t)bi * A.qstride;
    const float *__restrict__ c = A.c + (size_t)bi * A.cstride;

    const int j0 = (blockIdx.x * QG + qg) * QPW + lane;
    const int j1 = j0 + 64;
    const int j0c = min(j0, nq - 1), j1c = min(j1, nq - 1);
    f2 qx = {q[j0c * 3 + 0], q[j1c * 3 + 0]};
    f2 qy = {q[j0c * 3 + 1], q[j1c * 3 + 1]};
    f2 qz = {q[j0c * 3 + 2], q[j1c * 3 + 2]};
    if (DPF_NN_ABLATE & 8) { qx.x = lane * 0.001f; qx.y = qx.x + 0.1f; qy = qx * 0.5f; qz = qx * 0.25f; }

    // candidate slice of this wave, aligned to whole chunks
    const int nchunk = (nc + CH - 1) / CH;
    const int kbeg = (int)((long)nchunk * ks / KS) * CH;
    const int kend = min(nc, (int)((long)nchunk * (ks + 1) / KS) * CH);

    const float INF = __builtin_inff();
    f2 best = {INF, INF};
    int bc0 = kbeg, bc1 = kbeg;
    int k = kbeg;
    // Non-finite candidates (NaN / Inf) never show in a running minimum, but the reference's result depends on them
    // (nn_refscan.h).  The scan loop below is unrolled into groups of four iterations (64 candidates); at the top of a group
    // one 16-byte vector load per lane re-reads 256 of the slice's floats -- the scan itself uses scalar loads -- and at the
    // bottom the largest (bits << 1) is kept: an all-ones exponent is a value >= 0xFF000000.  Nothing is pending across
    // the loop's back-edge (a rotating pair of pending registers made the compiler wait vmcnt(0) at every latch: 7 % of the
    // scan at B = 32, r04), the load has ~2400 cycles to land, and the bookkeeping lives in VGPRs (`fi` = the float index of
    // this lane's next load; limits recomputed from kbeg / kend): the kernel sits at 78-79 SGPRs and a 256-thread block
    // loses a residency slot per CU above 80.
    uint32_t cexp = 0;
    int fi = kbeg * 3 + lane * 4;
    auto nf_load = [&]() -> u4a {
        u4a v = {0u, 0u, 0u, 0u};
        if (kend - kbeg >= 2) v = *(const u4a *)(c + min(fi, kend * 3 - 4));
        fi += 256;
        return v;
    };
    auto nf_fold = [&](u4a v) { cexp = max(max(cexp, v.x << 1), max(v.y << 1, max(v.z << 1, v.w << 1))); };
    const int nfull = (kend - kbeg) / CH;          // whole chunks in this slice
    if (nfull > 0 && !(DPF_NN_ABLATE & 4)) {
        // Software-pipelined scalar loads, ping-pong between two SGPR sets so
        // that chunk i+1's 24 floats are in flight while chunk i runs on the VALU.
        const int klast = kbeg + (nfull - 1) * CH;  // last whole chunk (prefetch clamp, never out of bounds)
        unsigned long long bufA[CH * 3 / 2], bufB[CH * 3 / 2];
#define DPF_LOAD_CHUNK(buf, kk)                                                  \
        {                                                                        \
            const unsigned long long *__restrict__ ck_ =                         \
                (const unsigned long long *)(c + (size_t)(kk) * 3); /* uniform, 8B-aligned (kk % 8 == 0) -> s_load */ \
            _Pragma("unroll") for (int u = 0; u < CH * 3 / 2; ++u) buf[u] = ck_[u]; \
        }
#define DPF_EVAL_HEAD(buf) f2 dm = pair_dist_sp<0>(buf, qx, qy, qz);
#define DPF_EVAL_TAIL(buf, kk)                                                   \
        chunk_min_sp<1>(buf, qx, qy, qz, dm);                                    \
        if (dm.x < best.x) { best.x = dm.x; bc0 = (kk); }                        \
        if (dm.y < best.y) { best.y = dm.y; bc1 = (kk); }
        // SMEM returns out of order, so any use of loaded SGPRs waits for ALL
        // outstanding scalar loads (lgkmcnt(0)).  Each prefetch is therefore
        // issued right AFTER the first use of the other buffer (its wait) and
        // has one whole chunk of VALU work to land.
        DPF_LOAD_CHUNK(bufA, k);
#define DPF_ITER2()                                                              \
        {                                                                        \
            {                                                                    \
                DPF_EVAL_HEAD(bufA);                                             \
                __builtin_amdgcn_sched_barrier(0);                               \
                DPF_LOAD_CHUNK(bufB, k + CH);                                    \
                __builtin_amdgcn_sched_barrier(0);                               \
                DPF_EVAL_TAIL(bufA, k);                                          \
            }                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                   \
            {                                                                    \
                DPF_EVAL_HEAD(bufB);                                             \
                __builtin_amdgcn_sched_barrier(0);                               \
                DPF_LOAD_CHUNK(bufA, min(k + 2 * CH, klast));                    \
                __builtin_amdgcn_sched_barrier(0);                               \
                DPF_EVAL_TAIL(bufB, k + CH);                                     \
            }                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                   \
            k += 2 * CH;                                                         \
        }
        int it = 0;
        if (!(DPF_NN_ABLATE & 1))
            for (; it + 8 <= nfull; it += 8) {       // 64 candidates scanned, 256 floats (85 candidates) checked
                const u4a v = nf_load();
                DPF_ITER2() DPF_ITER2() DPF_ITER2() DPF_ITER2()
                nf_fold(v);
            }
        for (; it + 2 <= nfull; it += 2) DPF_ITER2()
#undef DPF_ITER2
        if (it < nfull) {
            DPF_EVAL_HEAD(bufA);
            DPF_EVAL_TAIL(bufA, k);
            k += CH;
        }
#undef DPF_EVAL_HEAD
#undef DPF_EVAL_TAIL
#undef DPF_LOAD_CHUNK
    }
    if (k < kend) {   // ragged last chunk (only the last slice can have one)
        f2 dm = {INF, INF};
        for (int u = 0; k + u < kend; ++u) {
            const f2 d = pair_dist(c[(k + u) * 3 + 0], c[(k + u) * 3 + 1], c[(k + u) * 3 + 2], qx, qy, qz);
            dm.x = fminf(dm.x, d.x);
            dm.y = fminf(dm.y, d.y);
        }
        if (dm.x < best.x) { best.x = dm.x; bc0 = k; }
        if (dm.y < best.y) { best.y = dm.y; bc1 = k; }
    }

    if (kend > kbeg && !(DPF_NN_ABLATE & 1)) {  // the rows the main loop did not reach
        while (__any(fi - lane * 4 < kend * 3)) nf_fold(nf_load());
        if (kend - kbeg < 2)                   // a slice of one candidate: three floats, by hand
            for (int e = 0; e < 3; ++e) cexp = max(cexp, __builtin_bit_cast(uint32_t, c[kbeg * 3 + e]) << 1);
    }
    const bool cbad = __any(cexp >= 0xFF000000u);   // wave-uniform: this wave's candidate slice holds a NaN / Inf

    // recover the FIRST index inside the winning chunk (descending scan, last hit wins)
    int i0 = bc0, i1 = bc1;
    if (!(DPF_NN_ABLATE & 2))
#pragma unroll
    for (int u = CH - 1; u >= 0; --u) {
        const int k0 = min(bc0 + u, nc - 1), k1 = min(bc1 + u, nc - 1);
        const float d0 = one_dist(c[k0 * 3 + 0], c[k0 * 3 + 1], c[k0 * 3 + 2], qx.x, qy.x, qz.x);
        const float d1 = one_dist(c[k1 * 3 + 0], c[k1 * 3 + 1], c[k1 * 3 + 2], qx.y, qy.y, qz.y);
        if (d0 == best.x && bc0 + u < kend) i0 = bc0 + u;
        if (d1 == best.y && bc1 + u < kend) i1 = bc1 + u;
    }

    bool anybad = cbad;
    if constexpr (KS > 1) {
        __shared__ float sd[NWAVES][QPW];
        __shared__ int si[NWAVES][QPW];
        __shared__ int sbad[NWAVES];
        sd[wave][lane] = best.x; sd[wave][lane + 64] = best.y;
        si[wave][lane] = i0;     si[wave][lane + 64] = i1;
        if (lane == 0) sbad[wave] = cbad ? 1 : 0;
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int s = 1; s < KS; ++s) {   // ascending slices + strict '<' == global first minimum
            const float e0 = sd[wave + s][lane], e1 = sd[wave + s][lane + 64];
            if (e0 < best.x) { best.x = e0; i0 = si[wave + s][lane]; }
            if (e1 < best.y) { best.y = e1; i1 = si[wave + s][lane + 64]; }
            anybad = anybad || sbad[wave + s] != 0;
        }
    }
    // non-finite input: a NaN / Inf among the candidates (wave-uniform), or a query whose minimum never left +inf (its own
    // coordinates are NaN / Inf, or every distance overflowed): those queries get the reference's own scan (nn_refscan.h)
    if (anybad || nn_not_finite(best.x)) { float r; nn_reference_scan(c, nc, qx.x, qy.x, qz.x, r, i0); best.x = r; }
    if (anybad || nn_not_finite(best.y)) { float r; nn_reference_scan(c, nc, qx.y, qy.y, qz.y, r, i1); best.y = r; }
    if (j0 < nq) { A.dist[(size_t)bi * nq + j0] = best.x; A.idx[(size_t)bi * nq + j0] = i0; }
    if (j1 < nq) { A.dist[(size_t)bi * nq + j1] = best.y; A.idx[(size_t)bi * nq + j1] = i1; }
}

// ---- small problems: candidates staged in LDS, one query per lane (r04) ---------------------------------------------------
// A rank of an 8-GPU job holds 4-8 clouds of 2048 points: 16 384 - 32 768 queries per direction pair, 128 - 256 of nn_kernel's
// 128-query waves.  There the scan is bound by the LATENCY of its scalar loads (an L2-served s_load outlasts the ~300 cycles
// of VALU work one chunk of prefetch covers) and half the CUs have no workgroup.  This kernel is built for that regime:
//   * a workgroup = 64 queries (ONE per lane) x KSW candidate slices (one wave each), so 4 clouds give every CU a workgroup;
//   * the workgroup stages the cloud's candidates once, structure-of-arrays, in LDS (12 B per candidate; coalesced vector
//     loads; the NaN / Inf exponent test of nn_refscan.h rides on them); a chunk of 8 candidates is six broadcast
//     ds_read_b128 (every lane reads the same address), double-buffered in registers;
//   * the packed f32 math runs over PAIRS OF CANDIDATES against the lane's splatted query -- (c0x, c1x) - (qx, qx) ... --
//     4 packed instructions per candidate and 64 queries instead of 4.8, same formula, every operation rounded on its own;
//   * the rest is nn_kernel's: running minimum per chunk, first chunk that achieved it, one re-scan of that chunk for the
//     first index, slices merged in LDS in ascending order with strict '<', the reference's own loop for non-finite input.
// Bit-identical results (tests/test_gpu_chamfer.py runs every case through it: DPF_NN_SMALL / dpf_nn_small_mode).
template <int KSW>
__global__ __launch_bounds__(KSW * 64) void nn_small_kernel(NNArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds_c[];
    __shared__ float sd[KSW][64];
    __shared__ int si[KSW][64];
    __shared__ int sbad[KSW];
    const NNDir A = args.d[blockIdx.z];
    const int nq = A.nq, nc = A.nc;
    const int bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave 0 of every workgroup of the launch: publish the sum of this tile's distances; the cloud's last arriver adds the
    // 2 * gridDim.x sums in a fixed order (lane x takes tiles x, x + 64, ..; butterfly) and writes cd.  Payload and ticket
    // go through agent-scope accesses on both sides (the XCDs' L2s are not coherent with each other); tickets end at zero.
    auto publish = [&](float mine) {
        float t = mine;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
        const unsigned nwg2 = 2u * gridDim.x;
        float *p = args.part + (size_t)bi * nwg2;
        unsigned old = 0;
        if (lane == 0) {
            __hip_atomic_store(&p[blockIdx.z * gridDim.x + blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the sum has left this CU before the ticket is taken
            old = __hip_atomic_fetch_add(&args.ticket[bi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __shfl(old, 0);
        if (old != nwg2 - 1) return;
        float s1 = 0.f, s2 = 0.f;
        for (unsigned x = lane; x < gridDim.x; x += 64) {
            s1 += __hip_atomic_load(&p[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s2 += __hip_atomic_load(&p[gridDim.x + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
        if (lane == 0) {
            args.cd[bi] = s1 / (float)args.d[0].nq + s2 / (float)args.d[1].nq;
            __hip_atomic_store(&args.ticket[bi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if ((int)blockIdx.x * 64 >= nq) {
        if (args.part != nullptr && ks == 0) publish(0.f);
        return;
    }
    const float *__restrict__ q = A.q + (size_t)bi * A.qstride;
    const float *__restrict__ c = A.c + (size_t)bi * A.cstride;
    const int ncp = (nc + 7) & ~7;
    float *cx = lds_c, *cy = lds_c + ncp, *cz = lds_c + 2 * ncp;
    const int j = blockIdx.x * 64 + lane, jc = min(j, nq - 1);
    const float qx = q[jc * 3 + 0], qy = q[jc * 3 + 1], qz = q[jc * 3 + 2];
    uint32_t cexp = 0;
    constexpr int SB = 8;                                 // candidates per thread in flight: all loads of a batch, then the stores
    for (int base = 0; base < ncp; base += SB * KSW * 64) {
        float x[SB], y[SB], z[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {                    // padding: a point so far away that its distance is +inf
            const int k = base + u * KSW * 64 + tid, kc = min(k, nc - 1);
            x[u] = c[kc * 3 + 0]; y[u] = c[kc * 3 + 1]; z[u] = c[kc * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int k = base + u * KSW * 64 + tid;
            cexp = max(cexp, max(__builtin_bit_cast(uint32_t, x[u]) << 1, max(__builtin_bit_cast(uint32_t, y[u]) << 1, __builtin_bit_cast(uint32_t, z[u]) << 1)));
            if (k >= nc) x[u] = y[u] = z[u] = 3.0e38f;
            if (k < ncp) { cx[k] = x[u]; cy[k] = y[u]; cz[k] = z[u]; }
        }
    }
    const bool cbad = __any(cexp >= 0xFF000000u);
    if (lane == 0) sbad[ks] = cbad ? 1 : 0;
    __syncthreads();
    const int nchunk = ncp / CH;
    const int kbeg = (int)((long)nchunk * ks / KSW) * CH, kend = (int)((long)nchunk * (ks + 1) / KSW) * CH;
    const float INF = __builtin_inff();
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    float best = INF;
    int bc = kbeg;
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto chunk_min = [&](const f4 (&v)[6]) {              // v: x[0..3], x[4..7], y.., z..
        float dm = INF;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const f2 xa = {v[hlf][0], v[hlf][1]}, xb = {v[hlf][2], v[hlf][3]};
            const f2 ya = {v[2 + hlf][0], v[2 + hlf][1]}, yb = {v[2 + hlf][2], v[2 + hlf][3]};
            const f2 za = {v[4 + hlf][0], v[4 + hlf][1]}, zb = {v[4 + hlf][2], v[4 + hlf][3]};
            const f2 dxa = xa - qx2, dya = ya - qy2, dza = za - qz2;
            const f2 dxb = xb - qx2, dyb = yb - qy2, dzb = zb - qz2;
            const f2 da = (dxa * dxa + dya * dya) + dza * dza;
            const f2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
            dm = fminf(fminf(dm, da.x), da.y);
            dm = fminf(fminf(dm, db.x), db.y);
        }
        return dm;
    };
    auto load_chunk = [&](int k, f4 (&v)[6]) {
        v[0] = *(const f4 *)(cx + k); v[1] = *(const f4 *)(cx + k + 4);
        v[2] = *(const f4 *)(cy + k); v[3] = *(const f4 *)(cy + k + 4);
        v[4] = *(const f4 *)(cz + k); v[5] = *(const f4 *)(cz + k + 4);
    };
    if (kend > kbeg) {
        f4 va[6], vb[6];
        load_chunk(kbeg, va);
        int k = kbeg;
        for (; k + 2 * CH <= kend; k += 2 * CH) {
            load_chunk(k + CH, vb);
            const float d0 = chunk_min(va);
            if (d0 < best) { best = d0; bc = k; }
            load_chunk(min(k + 2 * CH, kend - CH), va);
            const float d1 = chunk_min(vb);
            if (d1 < best) { best = d1; bc = k + CH; }
        }
        if (k < kend) {
            const float d0 = chunk_min(va);
            if (d0 < best) { best = d0; bc = k; }
        }
    }
    // the FIRST index inside the winning chunk (descending scan, last hit wins); padding never matches a finite minimum.
    // (r05: the chunk comes back as six 16-byte reads, not 24 scalar ones -- the lanes' chunks differ, and 4-byte reads at
    // multiples of 8 floats use 8 of the 64 banks: r04 counters, SQ_LDS_BANK_CONFLICT = 0.21 of SQ_LDS_IDX_ACTIVE)
    int i0 = bc;
    {
        f4 w[6];
        load_chunk(bc, w);
#pragma unroll
        for (int u = CH - 1; u >= 0; --u) {
            const int kk = bc + u;
            const float d0 = one_dist(w[u >> 2][u & 3], w[2 + (u >> 2)][u & 3], w[4 + (u >> 2)][u & 3], qx, qy, qz);
            if (d0 == best && kk < nc) i0 = kk;
        }
    }
    sd[ks][lane] = best;
    si[ks][lane] = i0;
    __syncthreads();
    if (ks != 0) return;
    bool anybad = false;
#pragma unroll
    for (int s2 = 0; s2 < KSW; ++s2) anybad = anybad || sbad[s2] != 0;
#pragma unroll
    for (int s2 = 1; s2 < KSW; ++s2) {       // ascending slices + strict '<' == global first minimum
        const float e0 = sd[s2][lane];
        if (e0 < best) { best = e0; i0 = si[s2][lane]; }
    }
    if (anybad || nn_not_finite(best) || nn_query_far(qx, qy, qz)) nn_reference_scan(c, nc, qx, qy, qz, best, i0);
    if (j < nq) { A.dist[(size_t)bi * nq + j] = best; A.idx[(size_t)bi * nq + j] = i0; }
    if (args.part != nullptr) publish(j < nq ? best : 0.f);
}

// ---- backward -------------------------------------------------------------
// grad_xyz1[b,j] = 2*gd1[b,j]*(x1_j - x2[idx1_j])  -  sum_{l: idx2_l = j} 2*gd2[b,l]*(x2_l - x1_j)
// (nndistance.cu:139-145, both launches of :152-153).  The first term of each
// output has no collisions and is written with plain stores (which also
// replaces the reference's memset); only the scattered term uses atomics.
__global__ __launch_bounds__(256) void nn_grad_direct_kernel(int b, int n, const float *__restrict__ xyz1, int m,
                                                             const float *__restrict__ xyz2,
                                                             const float *__restrict__ gd1, const int *__restrict__ idx1,
                                                             const float *__restrict__ gd2, const int *__restrict__ idx2,
                                                             float *__restrict__ g1, float *__restrict__ g2) {
    const long total1 = (long)b * n, total = total1 + (long)b * m;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const bool first = t < total1;
        const long u = first ? t : t - total1;
        const int nn = first ? n : m, mm = first ? m : n;
        const float *xa = first ? xyz1 : xyz2, *xb = first ? xyz2 : xyz1;
        const float *gd = first ? gd1 : gd2;
        const int *idx = first ? idx1 : idx2;
        float *go = first ? g1 : g2;
        const long bi = u / nn;
        const int j2 = idx[u];
        const float g = gd[u] * 2;
        const float *pa = xa + u * 3, *pb = xb + (bi * mm + j2) * 3;
        go[u * 3 + 0] = g * (pa[0] - pb[0]);
        go[u * 3 + 1] = g * (pa[1] - pb[1]);
        go[u * 3 + 2] = g * (pa[2] - pb[2]);
    }
}

__global__ __launch_bounds__(256) void nn_grad_scatter_kernel(int b, int n, const float *__restrict__ xyz1, int m,
                                                              const float *__restrict__ xyz2,
                                                              const float *__restrict__ gd1, const int *__restrict__ idx1,
                                                              const float *__restrict__ gd2, const int *__restrict__ idx2,
                                                              float *g1, float *g2) {
    const long total1 = (long)b * n, total = total1 + (long)b * m;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const bool first = t < total1;
        const long u = first ? t : t - total1;
        const int nn = first ? n : m, mm = first ? m : n;
        const float *xa = first ? xyz1 : xyz2, *xb = first ? xyz2 : xyz1;
        const float *gd = first ? gd1 : gd2;
        const int *idx = first ? idx1 : idx2;
        float *go = first ? g2 : g1;          // the OTHER cloud's gradient
        const long bi = u / nn;
        const int j2 = idx[u];
        const float g = gd[u] * 2;
        const float *pa = xa + u * 3, *pb = xb + (bi * mm + j2) * 3;
        float *dst = go + (bi * mm + j2) * 3;
        atomicAdd(dst + 0, -(g * (pa[0] - pb[0])));
        atomicAdd(dst + 1, -(g * (pa[1] - pb[1])));
        atomicAdd(dst + 2, -(g * (pa[2] - pb[2])));
    }
}

// cd[b] = mean(dist1[b]) + mean(dist2[b])                          lib/networks/evaluating.py:112
// one 256-thread workgroup per cloud, every load of a cloud in flight at once, fixed-order tree
constexpr int RT = 256;
__global__ __launch_bounds__(RT) void chamfer_reduce_kernel(int n, int m, const float *__restrict__ d1,
                                                            const float *__restrict__ d2, float *__restrict__ cd) {
    __shared__ float red[2][RT / 64];
    const int bi = blockIdx.x, tid = threadIdx.x;
    const float *a = d1 + (size_t)bi * n, *b = d2 + (size_t)bi * m;
    float s1 = 0.f, s2 = 0.f;
    if (((n | m) & 3) == 0) {
        for (int j = tid * 4; j < n; j += 4 * RT) { const float4 v = *(const float4 *)(a + j); s1 += (v.x + v.y) + (v.z + v.w); }
        for (int j = tid * 4; j < m; j += 4 * RT) { const float4 v = *(const float4 *)(b + j); s2 += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int j = tid; j < n; j += RT) s1 += a[j];
        for (int j = tid; j < m; j += RT) s2 += b[j];
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0)
        cd[bi] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)n + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)m;
}

// f_score of lib/networks/utils.py:38-42 from the two distance rows of one nn_distance call:
//   precision = 100 * mean(dist2 < t), recall = 100 * mean(dist1 < t), F = 2 P R / (P + R + 1e-7)
// one workgroup per cloud, integer counts (exact, order-independent), the float arithmetic in the reference's order
__global__ __launch_bounds__(RT) void fscore_kernel(int n, int m, const float *__restrict__ d1, const float *__restrict__ d2,
                                                    float threshold, float *__restrict__ out) {
    __shared__ int red[2][RT / 64];
    const int bi = blockIdx.x, tid = threadIdx.x;
    const float *a = d1 + (size_t)bi * n, *b = d2 + (size_t)bi * m;
    int c1 = 0, c2 = 0;
    for (int j = tid; j < n; j += RT) c1 += a[j] < threshold ? 1 : 0;
    for (int j = tid; j < m; j += RT) c2 += b[j] < threshold ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = c1; red[1][tid >> 6] = c2; }
    __syncthreads();
    if (tid == 0) {
        const int r = red[0][0] + red[0][1] + red[0][2] + red[0][3], p = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float recall = 100.0f * ((float)r / (float)n), precision = 100.0f * ((float)p / (float)m);
        out[bi] = 2.0f * precision * recall / (precision + recall + 1e-7f);
    }
}

}  // namespace

// -1 (default): by size; 0: never; 1: whenever the clouds fit (tests / measurements).  env DPF_NN_SMALL.
static int g_nn_small_mode = getenv("DPF_NN_SMALL") ? atoi(getenv("DPF_NN_SMALL")) : -1;
extern "C" int dpf_nn_small_mode(int mode) {
    const int old = g_nn_small_mode;
    g_nn_small_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return old;
}

// rank-sized batches of mid-sized clouds: the LDS-staged kernel.  One workgroup per CU or fewer (B = 4 clouds of 2048 points:
// 10.4 us against the scalar-load scan's 14.9); with more the CUs that hold two workgroups set the pace and the scan that
// streams its candidates through SGPRs is as fast (B = 8: 16.0 vs 15.9 us; r04_small/sweep.txt).
bool nn_small_serves(int b, int n, int m) {       // (also asked by chamfer_mfma.hip's choice of kernel)
    const int nmax = n > m ? n : m, minc = n < m ? n : m;
    if (g_nn_small_mode == 0 || nmax > 8192 || b > 65535) return false;
    if (g_nn_small_mode == 1) return true;
    return minc >= 1024 && (long)((nmax + 63) / 64) * b * 2 <= 256;
}
static int nn_small_launch(const NNArgs &a, int b, int nmax, hipStream_t s) {
    const int lds = 3 * ((nmax + 7) & ~7) * (int)sizeof(float);
    // candidate slices per workgroup: enough waves for two per SIMD (a lone wave is bound by its own issue rate)
    static const int ksw_env = getenv("DPF_NN_KSW") ? atoi(getenv("DPF_NN_KSW")) : 0;
    const long wgs = (long)((nmax + 63) / 64) * b * 2;
    const int ksw = ksw_env ? ksw_env : (wgs * 4 >= 2048 ? 4 : 8);
    const dim3 grid((nmax + 63) / 64, b, 2);
    static LdsLimit limit4, limit8, limit16;
    if (ksw == 4) {
        if (hipError_t e = limit4.ensure((const void *)nn_small_kernel<4>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nn_small_kernel<4>, grid, dim3(4 * 64), lds, s, a);
    } else if (ksw == 8) {
        if (hipError_t e = limit8.ensure((const void *)nn_small_kernel<8>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nn_small_kernel<8>, grid, dim3(8 * 64), lds, s, a);
    } else {
        if (hipError_t e = limit16.ensure((const void *)nn_small_kernel<16>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nn_small_kernel<16>, grid, dim3(16 * 64), lds, s, a);
    }
    return (int)hipGetLastError();
}

// dpf_nndistance_cd's fast path for the same problems (chamfer_mfma.hip): DPF_ENOSUP when the kernel does not serve them.
// workspace layout as there: b tickets (zero on entry and on exit), then 2 * ceil(nmax / 64) sums per cloud.
int nn_small_cd(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i, float *result2,
                int *result2_i, float *cd, void *workspace, int tickets_are_zero, hipStream_t s) {
    if (!nn_small_serves(b, n, m)) return DPF_ENOSUP;
    if (!tickets_are_zero)
        if (hipError_t e = dpf_zero_async(workspace, (size_t)b * sizeof(unsigned), s); e != hipSuccess) return (int)e;
    NNArgs a;
    a.d[0] = NNDir{xyz, xyz2, result, result_i, n, m, (long)n * 3, (long)m * 3};
    a.d[1] = NNDir{xyz2, xyz, result2, result2_i, m, n, (long)m * 3, (long)n * 3};
    a.ticket = (unsigned *)workspace; a.part = (float *)workspace + b; a.cd = cd;
    return nn_small_launch(a, b, n > m ? n : m, s);
}

extern "C" int dpf_fscore_reduce(int b, int n, int m, const float *dist1, const float *dist2, float threshold, float *out,
                                 dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!dist1 || !dist2 || !out) return DPF_EINVAL;
    hipLaunchKernelGGL(fscore_kernel, dim3(b), dim3(RT), 0, (hipStream_t)stream, n, m, dist1, dist2, threshold, out);
    return (int)hipGetLastError();
}

extern "C" int dpf_chamfer_reduce(int b, int n, int m, const float *dist1, const float *dist2, float *cd,
                                  dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!dist1 || !dist2 || !cd) return DPF_EINVAL;
    hipLaunchKernelGGL(chamfer_reduce_kernel, dim3(b), dim3(RT), 0, (hipStream_t)stream, n, m, dist1, dist2, cd);
    return (int)hipGetLastError();
}

// nndistance with explicit per-cloud strides (in floats): stride 0 broadcasts ONE cloud against a
// whole batch, which is what pairwise_CD needs (lib/networks/utils.py:104-107 expands and copies
// cloud i N2 times before every call).
extern "C" int dpf_nndistance_strided(int b, int n, const float *xyz, long xyz_stride, int m, const float *xyz2,
                                      long xyz2_stride, float *result, int *result_i, float *result2, int *result2_i,
                                      dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0 || xyz_stride < 0 || xyz2_stride < 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    if ((long)n * 3 >= (1l << 31) || (long)m * 3 >= (1l << 31) || b > 65535) return DPF_ENOSUP;
    NNArgs a;
    a.d[0] = NNDir{xyz, xyz2, result, result_i, n, m, xyz_stride, xyz2_stride};     // nndistance.cu:126
    a.d[1] = NNDir{xyz2, xyz, result2, result2_i, m, n, xyz2_stride, xyz_stride};   // nndistance.cu:127
    const int nmax = n > m ? n : m;
    // pick the candidate split so that the launch has >= ~2 waves per SIMD on 256 CUs
    const long waves1 = (long)b * ((n + QPW - 1) / QPW + (m + QPW - 1) / QPW);
    hipStream_t s = (hipStream_t)stream;
    // Small problems (a rank's 4-8 clouds of 2048 points): with one wave per SIMD the scan is bound by the LATENCY of its
    // scalar loads (one chunk of prefetch covers ~300 cycles of VALU work, an L2-served s_load takes longer), so the
    // candidates are split over MORE waves -- 8 or 16 slices merged in LDS in ascending order -- until every SIMD has two
    static const int ks_env = getenv("DPF_NN_KS") ? atoi(getenv("DPF_NN_KS")) : 0;
    const int minc = n < m ? n : m;
    if (nn_small_serves(b, n, m) && !ks_env) return nn_small_launch(a, b, nmax, s);
    int ks_small = ks_env;
    if (!ks_small && waves1 < 512 && minc >= 1024) ks_small = 8;    // r04, B=4 N=2048: 4 slices 18.4 us, 8: 15.3, 16: 16.8
    if (ks_small == 16 || ks_small == 8) {
        dim3 grid((nmax + QPW - 1) / QPW, b, 2);
        if (ks_small == 16) hipLaunchKernelGGL((nn_kernel<16, 16>), grid, dim3(16 * 64), 0, s, a);
        else hipLaunchKernelGGL((nn_kernel<8, 8>), grid, dim3(8 * 64), 0, s, a);
        return (int)hipGetLastError();
    }
    if (waves1 >= 2048 || (n < 64 && m < 64)) {
        dim3 grid((nmax + NWAVES * QPW - 1) / (NWAVES * QPW), b, 2);
        hipLaunchKernelGGL(nn_kernel<1>, grid, dim3(NWAVES * 64), 0, s, a);
    } else if (waves1 >= 1024) {
        dim3 grid((nmax + 2 * QPW - 1) / (2 * QPW), b, 2);
        hipLaunchKernelGGL(nn_kernel<2>, grid, dim3(NWAVES * 64), 0, s, a);
    } else {
        dim3 grid((nmax + QPW - 1) / QPW, b, 2);
        hipLaunchKernelGGL(nn_kernel<4>, grid, dim3(NWAVES * 64), 0, s, a);
    }
    return (int)hipGetLastError();
}

extern "C" int dpf_nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                              float *result2, int *result2_i, dpf_stream_t stream) {
    return dpf_nndistance_strided(b, n, xyz, (long)n * 3, m, xyz2, (long)m * 3, result, result_i, result2, result2_i,
                                  stream);
}

extern "C" int dpf_nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1,
                                  const int *idx1, const float *grad_dist2, const int *idx2, float *grad_xyz1,
                                  float *grad_xyz2, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 || !grad_xyz2) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)b * (n + m);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(nn_grad_direct_kernel, dim3(blocks), dim3(256), 0, s, b, n, xyz1, m, xyz2, grad_dist1, idx1,
                       grad_dist2, idx2, grad_xyz1, grad_xyz2);
    hipLaunchKernelGGL(nn_grad_scatter_kernel, dim3(blocks), dim3(256), 0, s, b, n, xyz1, m, xyz2, grad_dist1, idx1,
                       grad_dist2, idx2, grad_xyz1, grad_xyz2);
    return (int)hipGetLastError();
}
