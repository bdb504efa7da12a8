// Approximate earth-mover's distance (soft auction matching) for gfx950.
//
// Replaces approxmatchkernel / matchcostkernel / matchcostgrad{1,2}kernel of the
// reference (lib/metrics/pytorch_structural_losses/src/approxmatch.cu:3-326).
//
// The reference runs ONE 512-thread block per cloud with block barriers
// between the 27 dependent passes (9 annealing levels x 3 passes), so B=16
// clouds use 16 of 256 CUs.  Here every pass is its own launch over
// (points x candidate-slices x clouds): a workgroup owns 64 points of one
// cloud, its waves split the inner loop over the other cloud (wave-uniform
// candidates come in through scalar loads), and partial sums merge in LDS in
// a fixed order.  The kernel boundary is the grid-wide barrier the algorithm
// needs between passes.  The (B, m, n) `match` tensor is the only large
// object: pass 3 read-modify-writes it once per level (the first level writes
// without reading, which replaces the reference's zero-fill, approxmatch.cu:16-17).
//
// Three forms, same entry points: (1) read-modify-write (no workspace): the reference's data flow, one launch per pass;
// (2) deferred, packed VALU (workspace): level state in the workspace, `match` written once, bit-identical to (1);
// (3) deferred, matrix cores (workspace, default since r05; further down): squared distances by MFMA, the passes over the
// LIVE points of cloud 2 only, tolerance parity with (1) / (2) -- which of (2) and (3) runs is decided per call on the device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "dpf_hip.h"
#include "zero_fill.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAXS = 16;   // max inner-loop slices (waves) per workgroup
constexpr int NLEVEL = 9;  // annealing levels j = 7 .. -1 (approxmatch.cu:24)

// Explicit fma / rn intrinsics: the read-modify-write and the deferred paths of approxmatch
// must produce the same bits, so nothing here is left to the compiler's contraction heuristics.
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ float fmul(float a, float b) { return a * b; }   // never contracted (file is contract(off))
__device__ __forceinline__ float fadd(float a, float b) { return a + b; }

__device__ __forceinline__ float bcast(float v, int lane) {   // lane wave-uniform -> v_readlane_b32 into an SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Stream the wave-uniform candidates [jb, je) of cloud Q (with NW per-candidate weight arrays
// w[0..NW), each indexed by candidate) past the lanes' own points: each lane loads ONE candidate
// of a 64-tile with coalesced vector loads (next tile in flight while the current one is used) and
// the tile is replayed by broadcasting lane u's registers (v_readlane -> SGPR operands).  No
// memory latency inside the 64-candidate loop, no LDS.
template <int NW, class F>
__device__ __forceinline__ void stream_candidates(const float *__restrict__ Q, const float *const (&w)[NW > 0 ? NW : 1],
                                                  int jb, int je, int lane, F &&f) {
    if (jb >= je) return;
    float cx, cy, cz, cw[NW > 0 ? NW : 1];
    auto load = [&](int j0, float &x, float &y, float &z, float (&ww)[NW > 0 ? NW : 1]) {
        const int jj = min(j0 + lane, je - 1);
        x = Q[jj * 3 + 0]; y = Q[jj * 3 + 1]; z = Q[jj * 3 + 2];
#pragma unroll
        for (int i = 0; i < NW; ++i) ww[i] = w[i][jj];
    };
    load(jb, cx, cy, cz, cw);
    for (int j0 = jb; j0 < je; j0 += 64) {
        float nx = 0.f, ny = 0.f, nz = 0.f, nw[NW > 0 ? NW : 1];
        if (j0 + 64 < je) load(j0 + 64, nx, ny, nz, nw);
        const int cnt = min(64, je - j0);
        if (cnt == 64) {
#pragma unroll 16
            for (int u = 0; u < 64; ++u) {
                float ww[NW > 0 ? NW : 1];
#pragma unroll
                for (int i = 0; i < NW; ++i) ww[i] = bcast(cw[i], u);
                f(j0 + u, bcast(cx, u), bcast(cy, u), bcast(cz, u), ww);
            }
        } else {
            for (int u = 0; u < cnt; ++u) {
                float ww[NW > 0 ? NW : 1];
#pragma unroll
                for (int i = 0; i < NW; ++i) ww[i] = bcast(cw[i], u);
                f(j0 + u, bcast(cx, u), bcast(cy, u), bcast(cz, u), ww);
            }
        }
        cx = nx; cy = ny; cz = nz;
#pragma unroll
        for (int i = 0; i < NW; ++i) cw[i] = nw[i];
    }
}

// Variant for kernels with no per-candidate weights: the wave-uniform candidates come through SCALAR loads
// (s_load -> SGPR operands), no v_readlane.  Measured r01: emd_grad1_kernel 2x faster with it, the ratio / match
// kernels (which also broadcast a weight per candidate and run at 8 waves/SIMD) 5 % slower.
template <class F>
__device__ __forceinline__ void stream_candidates_scalar(const float *__restrict__ Q, int jb, int je, F &&f) {
#pragma unroll 8
    for (int j = jb; j < je; ++j) f(j, Q[j * 3 + 0], Q[j * 3 + 1], Q[j * 3 + 2]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- deferred path: two points per lane, packed fp32 math, candidates as SGPR operands ----------------------
// Measured on gfx950 (tools/ubench/valu_rate.hip): a VALU instruction with an SGPR source issues at ~1.7 ns per
// wave (vs 1.0 ns for fp32 add/mul/fma on VGPRs only), v_readlane 1.75 ns, v_exp_f32 3.5 ns -- and a packed-f32
// instruction also 1.75 ns, SGPR source or not.  With ONE point per lane and the wave-uniform candidate in SGPRs a
// ratio step is 5 SGPR-source ops + 3 plain + exp = 16.8 ns per 64 pairs (measured, = the kernel's rate).  With TWO
// points per lane the same step is 8 v_pk_* + 2 exp = 21 ns per 128 pairs.  The candidates are (x, y, z, weight)
// records read by scalar loads as 64-bit pairs (x,y), (z,w); op_sel broadcasts one half of a pair to both packed
// lanes, so there is no SGPR->VGPR copy at all.  Every arithmetic step is the same IEEE operation as in the
// one-point kernels (v_pk_fma = fma per element), so the two paths stay bit-identical.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

template <int HALF>
__device__ __forceinline__ f2 bsub(u64 pr, f2 q) {   // {s,s} - q,  s = HALF-th float of the SGPR pair
    f2 r;
    if constexpr (HALF == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(pr), "v"(q));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(pr), "v"(q));
    return r;
}
template <int HALF>
__device__ __forceinline__ f2 bmul(u64 pr, f2 v) {   // {s,s} * v
    f2 r;
    if constexpr (HALF == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "s"(pr), "v"(v));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "s"(pr), "v"(v));
    return r;
}
__device__ __forceinline__ f2 bfma_hi(u64 pr, f2 e, f2 acc) {   // fma({s.hi,s.hi}, e, acc)
    f2 r;
    // `e` comes straight from v_exp_f32: a transcendental result needs one wait state before a VALU use, and the
    // compiler's hazard recogniser does not look into inline asm (without the s_nop the second packed lane read a
    // stale register)
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "s"(pr), "v"(e), "v"(acc));
    return r;
}
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 exp2_2(f2 x) { return f2{fast_exp2(x.x), fast_exp2(x.y)}; }
// |{q,q} - p|^2 with sqdist's association; q = (xy.lo, xy.hi, zw.lo)
__device__ __forceinline__ f2 sqdist2(f2 px, f2 py, f2 pz, u64 xy, u64 zw) {
    const f2 dx = bsub<0>(xy, px), dy = bsub<1>(xy, py), dz = bsub<0>(zw, pz);
    return fma2(dz, dz, fma2(dy, dy, dx * dx));
}

// Stream records of NP 64-bit pairs each, [jb, je) of C, through scalar loads: chunks of CHK records ping-pong
// between two SGPR sets; scalar loads return out of order (any use waits for ALL outstanding ones), so the next
// chunk is requested right after the first use of the current one and has the rest of it to land.
template <int NP, int CHK, class F>
__device__ __forceinline__ void stream_records(const u64 *__restrict__ C, int jb, int je, F &&f) {
    constexpr int W = NP * CHK;
    int j = jb;
    const int nfull = (je - jb) / CHK;
    auto use = [&](const u64 (&buf)[W], int jj, int u) { f(jj + u, &buf[u * NP]); };
    if (nfull > 0) {
        const int jlast = jb + (nfull - 1) * CHK;
        u64 bufA[W], bufB[W];
#define RC_LOAD(buf, jj)                                                                   \
        {                                                                                  \
            const u64 *__restrict__ cp_ = C + (size_t)(jj) * NP;   /* wave-uniform */      \
            _Pragma("unroll") for (int u = 0; u < W; ++u) buf[u] = cp_[u];                 \
        }
#define RC_REST(buf, jj) _Pragma("unroll") for (int u = 1; u < CHK; ++u) use(buf, jj, u);
        RC_LOAD(bufA, j);
        int it = 0;
        for (; it + 2 <= nfull; it += 2, j += 2 * CHK) {
            use(bufA, j, 0);
            __builtin_amdgcn_sched_barrier(0);
            RC_LOAD(bufB, j + CHK);
            __builtin_amdgcn_sched_barrier(0);
            RC_REST(bufA, j)
            __builtin_amdgcn_sched_barrier(0);
            use(bufB, j + CHK, 0);
            __builtin_amdgcn_sched_barrier(0);
            RC_LOAD(bufA, min(j + 2 * CHK, jlast));
            __builtin_amdgcn_sched_barrier(0);
            RC_REST(bufB, j + CHK)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it < nfull) {
            use(bufA, j, 0);
            RC_REST(bufA, j)
            j += CHK;
        }
#undef RC_REST
#undef RC_LOAD
    }
    for (; j < je; ++j) {
        u64 one[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) one[u] = C[(size_t)j * NP + u];
        f(j, &one[0]);
    }
}

constexpr int PPW = 128;   // points per wave in the deferred kernels

// (xyz2, multiR) records for the first ratio pass
// `gate`: the deferred path comes in two families -- these packed-VALU kernels (difference-form d^2, bit-identical to the
// read-modify-write path) and the matrix-core passes further down (expanded-form d^2).  Which one runs is decided ON THE DEVICE
// (emd_mfma_prep_kernel looks at the coordinates' range), so both families are launched and every kernel of the family that is
// not wanted returns at once: gate == nullptr -> always run, else run iff *gate == want.
__device__ __forceinline__ bool gate_closed(const unsigned *gate, unsigned want) {
    return gate != nullptr && __builtin_nontemporal_load(gate) != want;
}

__global__ void emd_pack_init_kernel(int m, float multiR, const float *__restrict__ xyz2, float4 *__restrict__ c2a, size_t pstride,
                                     const unsigned *gate) {
    if (gate_closed(gate, 1u)) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float *q = xyz2 + ((size_t)blockIdx.y * m + j) * 3;
    c2a[(size_t)blockIdx.y * pstride + j] = make_float4(q[0], q[1], q[2], multiR);
}

// remainL = multiL, remainR = multiR                       approxmatch.cu:6-12,18-21
__global__ void emd_init_kernel(int n, int m, float multiL, float multiR, float *__restrict__ temp, unsigned *flag) {
    if (flag != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *flag = 0u;
    float *t = temp + (size_t)blockIdx.y * (n + m) * 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x)
        t[i] = i < n ? multiL : multiR;
}

// Passes 1 and 2 share one shape: for every point i of cloud P
//     s_i = sum_j exp(level * |P_i - Q_j|^2) * wq[j]
// PASS 1 (P=xyz1, Q=xyz2, wq=remainR): ratioL[i] = remainL[i] / (1e-9 + s_i)    approxmatch.cu:29-62
// PASS 2 (P=xyz2, Q=xyz1, wq=ratioL):  sumr = s_i * remainR[i];                  approxmatch.cu:78-111
//        ratioR[i] = min(remainR[i]/(sumr+1e-9), 1) * remainR[i]; remainR[i] = max(0, remainR[i]-sumr)
// ratio vectors live at rbase + bi*rstride: [ratioL(n) | ratioR(m)] -- either inside `temp`
// (reference layout, approxmatch.cu:4) or in a per-level slot of the caller's workspace
template <int PASS>
__global__ __launch_bounds__(1024) void emd_ratio_kernel(int n, int m, float lvl2, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2, float *temp, float *rbase,
                                                         size_t rstride) {
    __shared__ float part[MAXS][64];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    float *t = temp + (size_t)bi * (n + m) * 2;
    float *remainL = t, *remainR = t + n, *ratioL = rbase + (size_t)bi * rstride, *ratioR = ratioL + n;
    const int np = PASS == 1 ? n : m, nq = PASS == 1 ? m : n;
    const float *__restrict__ P = (PASS == 1 ? xyz1 + (size_t)bi * n * 3 : xyz2 + (size_t)bi * m * 3);
    const float *__restrict__ Q = (PASS == 1 ? xyz2 + (size_t)bi * m * 3 : xyz1 + (size_t)bi * n * 3);
    const float *__restrict__ wq = PASS == 1 ? remainR : ratioL;
    const int i = blockIdx.x * 64 + lane;
    const int ic = min(i, np - 1);
    const float px = P[ic * 3 + 0], py = P[ic * 3 + 1], pz = P[ic * 3 + 2];
    const int jb = (int)((long)nq * slice / S), je = (int)((long)nq * (slice + 1) / S);
    float s = 0.f;
    const float *const wl[1] = {wq};
    stream_candidates<1>(Q, wl, jb, je, lane, [&](int, float qx, float qy, float qz, const float (&ww)[1]) {
        s = __builtin_fmaf(fast_exp2(fmul(lvl2, sqdist(px, py, pz, qx, qy, qz))), ww[0], s);
    });
    part[slice][lane] = s;
    __syncthreads();
    if (slice != 0 || i >= np) return;
    float tot = PASS == 1 ? 1e-9f : 0.f;
    for (int u = 0; u < S; ++u) tot += part[u][lane];
    if (PASS == 1) {
        ratioL[i] = remainL[i] / tot;
    } else {
        const float rr = remainR[i];
        const float sumr = tot * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        ratioR[i] = consumption * rr;
        remainR[i] = fmaxf(0.0f, rr - sumr);
    }
}

// PASS 3: w = exp(level*d^2) * ratioL[k] * ratioR[l]; match[l][k] += w;          approxmatch.cu:130-163
//         remainL[k] = max(0, remainL[k] - sum_l w)
// MODE 0: match = w (first level), 1: match += w, 2: no match traffic (deferred materialisation)
template <int MODE>
__global__ __launch_bounds__(1024) void emd_match_kernel(int n, int m, float lvl2, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2, float *__restrict__ match,
                                                         float *temp, const float *rbase, size_t rstride) {
    constexpr bool FIRST = MODE == 0;
    __shared__ float part[MAXS][64];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    float *t = temp + (size_t)bi * (n + m) * 2;
    float *remainL = t;
    const float *__restrict__ ratioL = rbase + (size_t)bi * rstride;
    const float *__restrict__ ratioR = ratioL + n;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + (size_t)bi * m * 3;
    float *__restrict__ mt = match + (size_t)bi * n * m;
    const int k = blockIdx.x * 64 + lane;
    const bool live = k < n;
    const int kc = min(k, n - 1);
    const float px = P[kc * 3 + 0], py = P[kc * 3 + 1], pz = P[kc * 3 + 2];
    const float rl = ratioL[kc];
    const int lb = (int)((long)m * slice / S), le = (int)((long)m * (slice + 1) / S);
    float suml = 0.f;
    const float *const wl[1] = {ratioR};
    stream_candidates<1>(Q, wl, lb, le, lane, [&](int l, float qx, float qy, float qz, const float (&ww)[1]) {
        const float w = fmul(fmul(fast_exp2(fmul(lvl2, sqdist(px, py, pz, qx, qy, qz))), rl), ww[0]);
        if (MODE != 2 && live) {
            float *dst = mt + (size_t)l * n + k;
            *dst = FIRST ? w : fadd(*dst, w);
        }
        suml = fadd(suml, w);
    });
    part[slice][lane] = suml;
    __syncthreads();
    if (slice != 0 || !live) return;
    float tot = 0.f;
    for (int u = 0; u < S; ++u) tot += part[u][lane];
    remainL[k] = fmaxf(0.0f, remainL[k] - tot);
}

// Deferred materialisation: match[l][k] = sum over the 9 levels (in level order, so the fp32
// association equals the reference's repeated `match += w`, approxmatch.cu:155) of
// exp(level*d^2) * ratioL_level[k] * ratioR_level[l].  One 4-byte write per pair instead of a
// read-modify-write per level: HBM traffic 4*n*m instead of 68*n*m bytes per cloud.
struct Levels { float lvl2[NLEVEL]; };

// ---- the deferred path's kernels (two points per lane; see the note above bsub) -------------------------------
// Packed records per cloud, pk + bi*pstride float4's: [C1 (n) | C2a (m) | C2b (m)] =
// (xyz1, ratioL) | (xyz2, remainR) | (xyz2, ratioR); each pass writes the record the next pass streams.
// Same slices and the same per-lane candidate order as emd_ratio_kernel / emd_match_kernel => same bits.
template <int PASS>
__global__ __launch_bounds__(1024) void emd_ratio2_kernel(int n, int m, float lvl2, const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2, float *temp, float *rbase,
                                                          size_t rstride, float4 *pk, size_t pstride, const unsigned *gate) {
    __shared__ float part[MAXS][PPW];
    if (gate_closed(gate, 1u)) return;
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    float *t = temp + (size_t)bi * (n + m) * 2;
    float *remainL = t, *remainR = t + n, *ratioL = rbase + (size_t)bi * rstride, *ratioR = ratioL + n;
    const int np = PASS == 1 ? n : m, nq = PASS == 1 ? m : n;
    const float *__restrict__ P = (PASS == 1 ? xyz1 + (size_t)bi * n * 3 : xyz2 + (size_t)bi * m * 3);
    float4 *c1 = pk + (size_t)bi * pstride, *c2a = c1 + n, *c2b = c2a + m;
    const int i0 = blockIdx.x * PPW + lane, i1 = i0 + 64;
    const int a0 = min(i0, np - 1), a1 = min(i1, np - 1);
    const f2 px = {P[a0 * 3 + 0], P[a1 * 3 + 0]}, py = {P[a0 * 3 + 1], P[a1 * 3 + 1]}, pz = {P[a0 * 3 + 2], P[a1 * 3 + 2]};
    const int jb = (int)((long)nq * slice / S), je = (int)((long)nq * (slice + 1) / S);
    const u64 lv = ((u64)__float_as_uint(lvl2) << 32) | __float_as_uint(lvl2);
    f2 s = {0.f, 0.f};
    stream_records<2, 8>((const u64 *)(PASS == 1 ? c2a : c1), jb, je, [&](int, const u64 *r) {
        const f2 e = exp2_2(bmul<0>(lv, sqdist2(px, py, pz, r[0], r[1])));
        s = bfma_hi(r[1], e, s);
    });
    part[slice][lane] = s.x;
    part[slice][lane + 64] = s.y;
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = h ? i1 : i0;
        if (i >= np) continue;
        const float qx = h ? px.y : px.x, qy = h ? py.y : py.x, qz = h ? pz.y : pz.x;
        float tot = PASS == 1 ? 1e-9f : 0.f;
        for (int u = 0; u < S; ++u) tot += part[u][lane + 64 * h];
        if (PASS == 1) {
            const float r = remainL[i] / tot;
            ratioL[i] = r;
            c1[i] = make_float4(qx, qy, qz, r);
        } else {
            const float rr = remainR[i];
            const float sumr = tot * rr;
            const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
            const float r = consumption * rr, rem = fmaxf(0.0f, rr - sumr);
            ratioR[i] = r;
            remainR[i] = rem;
            c2b[i] = make_float4(qx, qy, qz, r);
            c2a[i] = make_float4(qx, qy, qz, rem);
        }
    }
}

__global__ __launch_bounds__(1024) void emd_match2_kernel(int n, int m, float lvl2, const float *__restrict__ xyz1,
                                                          float *temp, const float *rbase, size_t rstride,
                                                          const float4 *pk, size_t pstride, const unsigned *gate) {
    __shared__ float part[MAXS][PPW];
    if (gate_closed(gate, 1u)) return;
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    float *remainL = temp + (size_t)bi * (n + m) * 2;
    const float *__restrict__ ratioL = rbase + (size_t)bi * rstride;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const int k0 = blockIdx.x * PPW + lane, k1 = k0 + 64;
    const int a0 = min(k0, n - 1), a1 = min(k1, n - 1);
    const f2 px = {P[a0 * 3 + 0], P[a1 * 3 + 0]}, py = {P[a0 * 3 + 1], P[a1 * 3 + 1]}, pz = {P[a0 * 3 + 2], P[a1 * 3 + 2]};
    const f2 rl = {ratioL[a0], ratioL[a1]};
    const int lb = (int)((long)m * slice / S), le = (int)((long)m * (slice + 1) / S);
    const u64 lv = ((u64)__float_as_uint(lvl2) << 32) | __float_as_uint(lvl2);
    f2 suml = {0.f, 0.f};
    stream_records<2, 8>((const u64 *)(pk + (size_t)bi * pstride + n + m), lb, le, [&](int, const u64 *r) {
        const f2 e = exp2_2(bmul<0>(lv, sqdist2(px, py, pz, r[0], r[1])));
        suml = suml + bmul<1>(r[1], e * rl);
    });
    part[slice][lane] = suml.x;
    part[slice][lane + 64] = suml.y;
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = h ? k1 : k0;
        if (k >= n) continue;
        float tot = 0.f;
        for (int u = 0; u < S; ++u) tot += part[u][lane + 64 * h];
        remainL[k] = fmaxf(0.0f, remainL[k] - tot);
    }
}


// ---- the deferred path on the matrix cores (r05; operand format r06) ------------------------------------------------------
// Six of the eight packed instructions per candidate above are the squared distance.  Expanded, -4^j log2(e) |q - p|^2 is a
// contraction over K slots and the matrix cores hand a wave the exp2 arguments of 32 x 32 pairs at once; what is left per pair
// is exp2 and one FMA with the candidate's weight -- 32 VALU instructions per 1024 pairs where the packed form issues 80.
//
// r06 -- the operand format (the r05 format, fp16 hi + lo of the coordinates against fp16 parts of the squared
// norms in ONE MFMA, left the exponent with an rms error of 5e-4 and a worst case of 4e-3 at the steepest level: the three
// large terms -s|q|^2 - s|p|^2 + 2s q.p, each in the thousands, cancel INSIDE the fp32 accumulator, which aligns every 9-input
// partial sum to its largest term (profiles/r05_mfma_tree.txt, r05_mfma_accum.txt).  The auction averages that out on generic
// clouds -- cost within 2e-5 of the difference form -- but not on degenerate ones: 1.15e-4 on collinear points,
// tests/diag/emd_collinear_case.py, outside the 1e-4 contract).  Now the large terms are EXACT:
//   q' = sqrt(log2 e) (q - c) 2^g = a + alpha,   a = rint(q') an integer vector with |a_u| <= 1000, |alpha_u| <= 1/2
// (c: cloud 1's centroid; g per cloud pair from the clouds' extent, <= 11), likewise p' = b + beta for cloud 1, and
//   -|q' - p'|^2 = [ -|a|^2 - |b|^2 + 2 a.b ]  +  [ -R_q - C_p + 2 a.beta + 2 alpha.b ]  +  2 alpha.beta,
//   R_q = 2 a.alpha + |alpha|^2,  C_p = 2 b.beta + |beta|^2.
// The first bracket is integer arithmetic on numbers below 2^22: fp16 holds every operand exactly (|a|^2 as two 11-bit pieces),
// fp32 accumulation is exact whatever the order -- and the instruction adds the products of K slots 0..7 (with C) BEFORE those
// of slots 8..15 (the tree test above), so the integers go into slots 0..7 of the first MFMA and have cancelled to
// -|a - b|^2 (small for every pair whose weight matters) before anything inexact is added.  The second bracket -- terms up to
// 3000 that cancel to 2 (a - b).(alpha - beta) -- takes slots 8..15 with the hi parts of R, C, alpha, beta and, in a second
// MFMA chained through C, the lo parts (fp16 hi + lo = 22 bits) and alpha_h.beta_h.  Per 32 x 32 pairs: two MFMAs (the matrix
// pipe was a tenth busy), the same 32 VALU instructions.  Exponent error at the steepest level on unit-size clouds: max
// 1e-5, rms 2e-6 -- the class of the fp32 difference form's own rounding (5e-6); measured on the device by
// test_matrix_core_exponent_error_bound through dpf_debug_emd_exponents, emulated on the CPU by tests/diag/emd_grid_emulation.py.
// Scales: the records carry the steepest level's 4^7 2^-2g = 2^T (T = 14 - 2g in [-8, 0]) split per slot so that an INTEGER
// operand is never below fp16's smallest normal number (integers lost to a flush would break the exact cancellation) and a
// small operand's hi part never is; lo parts may underflow -- gradually where the hardware honours fp16 subnormals, to zero
// where it does not: either way below 8e-6 of the exponent (slot table at point_records).  Level j's 4^(j-7) goes onto the
// fragments as a per-slot vector of powers of two (level_vectors): onto the integers of an all-integer slot, onto the small
// operand of a mixed one -- 4 exact v_pk_mul_f16 per fragment, as before.
// The results are NOT bit-identical to the read-modify-write path (tolerance parity; the same weight bits in all three passes of
// a level and in the materialisation, below); the packed-VALU family above still is and stays the path for clouds whose extent
// puts the second bracket's rounding above 1e-4 (log2(e) |x - c|^2 > 16 for some point, or not finite), chosen per call on
// the device (gate), and for dpf_emd_set_matrix_path(0).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int MT = 4;            // 32-point tiles per wave
constexpr int MPW = 32 * MT;     // points per wave / workgroup
constexpr int MSL = 8;           // max candidate slices (waves) per workgroup
constexpr float EMD_MFMA_R2MAX = 16.f;    // log2(e) |x - c|^2 bound: g >= 7, T <= 0, the mixed terms' rounding <= 2e-4 at level 7
constexpr float EMD_AMAX = 1000.f;        // bound of the integer parts (2 a and |a|^2 / 2048 stay 11-bit integers)
constexpr int EMD_GMAX = 11;              // finest grid 2^-11: T >= -8
constexpr int RECQ = 4;                   // 16-byte quarters per point record: [MFMA 1: K 0..7 | K 8..15 | MFMA 2: K 0..7 | K 8..15]
constexpr int METAF = 8;                  // floats of per-cloud meta: centroid (3), out-of-range flag, 2^g, T

__device__ __forceinline__ int round_up(int v, int q) { return (v + q - 1) / q * q; }

// One workgroup per cloud: cloud 1's centroid (fixed-order sums), the grid exponent g from the largest centred, scaled
// coordinate of either cloud, and whether every point of both clouds is in range
__global__ __launch_bounds__(1024) void emd_mfma_prep_kernel(int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                             float *__restrict__ meta, unsigned *flag) {
    __shared__ float red[3][1024];
    const int bi = blockIdx.x, tid = threadIdx.x;
    const float *p1 = xyz1 + (size_t)bi * n * 3, *p2 = xyz2 + (size_t)bi * m * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = tid; i < n; i += 1024) { sx += p1[i * 3 + 0]; sy += p1[i * 3 + 1]; sz += p1[i * 3 + 2]; }
    red[0][tid] = sx; red[1][tid] = sy; red[2][tid] = sz;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
        __syncthreads();
    }
    const float cx = red[0][0] / (float)n, cy = red[1][0] / (float)n, cz = red[2][0] / (float)n;
    __syncthreads();
    int bad = 0;
    float mx = 0.f;
    for (int i = tid; i < n + m; i += 1024) {
        const float *q = i < n ? p1 + i * 3 : p2 + (i - n) * 3;
        const float dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz;
        const float r2 = 1.44269504f * (dx * dx + dy * dy + dz * dz);
        bad |= !(r2 <= EMD_MFMA_R2MAX);                 // also catches NaN / inf
        mx = fmaxf(mx, fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz))));     // (fmaxf drops a NaN: `bad` has it)
    }
    red[0][tid] = mx;
    bad = __syncthreads_or(bad);
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) red[0][tid] = fmaxf(red[0][tid], red[0][tid + o]);
        __syncthreads();
    }
    if (tid == 0) {
        // 2^g: the largest power of two with 1.2012 mx 2^g <= 1000 (1.2012 > sqrt(log2 e): the records' double arithmetic stays
        // inside); clouds of no extent take the finest grid
        const float lim = EMD_AMAX / (1.2012f * fmaxf(red[0][0], 1e-30f));
        int g = bad ? 7 : (lim >= 4096.f ? EMD_GMAX : min(EMD_GMAX, (int)floorf(log2f(lim))));
        while (g > 0 && ldexpf(1.2012f * red[0][0], g) > EMD_AMAX) --g;              // (log2f's last bit)
        float *mt = meta + (size_t)bi * METAF;
        mt[0] = cx; mt[1] = cy; mt[2] = cz; mt[3] = (float)bad; mt[4] = ldexpf(1.f, g); mt[5] = (float)(14 - 2 * g);
        mt[6] = 0.f; mt[7] = 0.f;
        if (bad) atomicOr(flag, 1u);
    }
}

__device__ __forceinline__ void split2(double v, _Float16 &h, _Float16 &l) {
    h = (_Float16)(float)v;
    l = (_Float16)(float)(v - (double)(float)h);
}

// Sparsity (r05, tests/diag/emd_level_zeros.py): a point of cloud 2 whose remainR has reached 0 stays at 0 -- its weight in
// pass 1 (remainR) and pass 3 (ratioR) is exactly 0 and pass 2 has nothing to compute for it.  On bench.py's cfg5 clouds the
// live fraction per level is 1, 0.58, 0.31, 0.13, 0.05, 0.02, 0.007, 0.001, 0 (independent clouds: 1, 0.59, 0.35, 0.20,
// 0.11, 0.07, 0.03, 0.007, 0.002): 2.1 - 2.4 level-passes' worth of pairs instead of 9.  The passes therefore walk a COMPACTED
// list of cloud 2's live points (A records, original indices, remainR; ping-pong buffers, one stable compaction per level:
// exact zeros dropped, so the sums lose only terms that are +0), and the compaction files every point it drops into `order`
// behind the points that outlive it: in that order every level's list is a prefix, and the materialisation gives row tile T
// only the levels whose list reaches it.
struct MfmaState {
    int n, m, NP, MP, nb;
    const u4 *recB1;        // [nb][NP] B records of cloud 1 (dense, padded with zero records)
    u4 *recA2[2];           // [nb][MP] A records of cloud 2's live points
    int *idx2[2];           //          their original indices
    float *remainR_c[2];    //          their remainR
    float *ratioR_c;        // [nb][MP] their remainR - ratioR of the current level (pass 3's row weights; r05: ratioR)
    float *ratioL_p;        // [nb][NP] cloud 1's ratioL of the current level
    int *count;             // [nb] length of the current list
    int *counts;            // [NLEVEL][nb] length of every level's list
    int *order;             // [nb][MP] cloud 2's points, longest-lived first
    float *temp;            // reference layout: remainL (n) | remainR (m) | ...
    size_t rstride;         // per-cloud stride of a level's ratio slot [ratioL (n) | ratioR (m)]
    const unsigned *gate;
};

// fp16 operand records of a point (header of this section).  In grid units q' = a + alpha (rows: cloud 2, the A operand),
// p' = b + beta (columns: cloud 1, the B operand); every product below times 2^T 4^(j-7) sums to -4^j log2(e) |q - p|^2.
//   slot            A operand (x its scale)              B operand (x its scale)            level factor goes onto
//   MFMA 1, K 0..7  -N_hi, -N_lo, -2048, -1, 2 a_u  (2^4)  2048, 1, N_hi, N_lo, b_u (2^(T-4))   A (integers stay >= 2^-12)
//   MFMA 1, K 8     -hi(8 R_q)                            2^(T-3)                              A
//           K 9     -2^(T-3)                              hi(8 C_p)                            B
//           K 10-12 2 a_u            (2^-14)              hi(beta_u 2^(T+14))                  B
//           K 13-15 hi(alpha_u 2^(T+14))                  2 b_u            (2^-14)             A
//   MFMA 2, K 0..7  the same eight slots with the lo parts of R_q, C_p, beta_u, alpha_u
//   MFMA 2, K 8-10  2 alpha_u 2^(T/2)                     beta_u 2^(T/2)                       both (2^(j-7) each)
// |a|^2 = 2048 N_hi + N_lo (both below 2048).  An integer operand is >= 2^-14 (normal) at every level but one corner (a_u = 1 at
// the last level, a term below 2e-5); a lo part underflows where it is below 2^-14 of its slot's scale: < 8e-6 of the exponent.
__device__ __forceinline__ void point_records(const float *q, const float *mt, bool live, u4 (&ra)[RECQ], u4 (&rb)[RECQ]) {
    _Float16 a[32], b[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { a[u] = (_Float16)0.f; b[u] = (_Float16)0.f; }
    if (live) {
        const double S = 1.2011224087864498;                      // sqrt(log2 e)
        const double G = (double)mt[4];
        const int T = (int)mt[5];
        const double sI = ldexp(1.0, T - 4), sC = ldexp(1.0, T - 3), sS = ldexp(1.0, T + 14), sH = ldexp(1.0, T / 2);
        double ip[3], fp[3], N = 0.0, R = 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const double v = ((double)q[u] - (double)mt[u]) * S * G;
            ip[u] = rint(v); fp[u] = v - ip[u];
            N += ip[u] * ip[u];
            R += 2.0 * ip[u] * fp[u] + fp[u] * fp[u];
        }
        const double Nh = floor(N / 2048.0), Nl = N - 2048.0 * Nh;
        // rows (A)
        a[0] = (_Float16)(float)(-16.0 * Nh); a[1] = (_Float16)(float)(-16.0 * Nl); a[2] = (_Float16)(-32768.f); a[3] = (_Float16)(-16.f);
        split2(-8.0 * R, a[8], a[16]);
        a[9] = a[17] = (_Float16)(float)(-sC);
        // columns (B)
        b[0] = (_Float16)(float)(2048.0 * sI); b[1] = (_Float16)(float)sI; b[2] = (_Float16)(float)(Nh * sI); b[3] = (_Float16)(float)(Nl * sI);
        b[8] = b[16] = (_Float16)(float)sC;
        split2(8.0 * R, b[9], b[17]);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            a[4 + u] = (_Float16)(float)(32.0 * ip[u]);
            a[10 + u] = a[18 + u] = (_Float16)(float)ldexp(2.0 * ip[u], -14);
            split2(fp[u] * sS, a[13 + u], a[21 + u]);
            a[24 + u] = (_Float16)(float)(2.0 * fp[u] * sH);
            b[4 + u] = (_Float16)(float)(ip[u] * sI);
            split2(fp[u] * sS, b[10 + u], b[18 + u]);
            b[13 + u] = b[21 + u] = (_Float16)(float)ldexp(2.0 * ip[u], -14);
            b[24 + u] = (_Float16)(float)(fp[u] * sH);
        }
    }
    __builtin_memcpy(ra, a, 64);
    __builtin_memcpy(rb, b, 64);
}

// Level j's 4^(j-7) as per-slot powers of two.  F = max(4^(j-7), 2^-14) goes onto the operand the table above names, f = 4^(j-7) / F
// (1 but for the last level's 2^-2) onto the other, h = 2^(j-7) onto both operands of the alpha.beta slots.  fp16 bit patterns.
struct LevelFac { unsigned F, f, h; };
__device__ __forceinline__ unsigned pk2(unsigned lo, unsigned hi) { return lo | (hi << 16); }
// the two quarters a lane of half `half` multiplies its fragments by: [0] MFMA 1, [1] MFMA 2
__device__ __forceinline__ void level_vectors(LevelFac lf, int half, bool rows, u4 (&v)[2]) {
    const unsigned own = rows ? lf.F : lf.f, oth = rows ? lf.f : lf.F;     // `own`: this side's factor where A takes the level
    const u4 ints = {pk2(own, own), pk2(own, own), pk2(own, own), pk2(own, own)};
    const u4 mixed = {pk2(own, oth), pk2(oth, oth), pk2(oth, own), pk2(own, own)};
    const u4 both = {pk2(lf.h, lf.h), pk2(lf.h, lf.h), pk2(lf.h, lf.h), pk2(lf.h, lf.h)};
    v[0] = half == 0 ? ints : mixed;
    v[1] = half == 0 ? mixed : both;
}

// B records of cloud 1, A records of cloud 2 (dense copy + the first list: every point, in order), zeros in the padding,
// zeros in every level's ratioR slot of the workspace (a point that has left the auction is not written again)
__global__ void emd_mfma_pack_kernel(MfmaState st, float multiR, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                     const float *__restrict__ meta, u4 *__restrict__ recA2_dense, float *__restrict__ ws, size_t lstride) {
    if (gate_closed(st.gate, 0u)) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
    if (idx >= st.NP + st.MP) return;
    const bool first = idx < st.NP;
    const int k = first ? idx : idx - st.NP;
    const bool live = k < (first ? st.n : st.m);
    const float *q = first ? xyz1 + ((size_t)bi * st.n + (live ? k : 0)) * 3 : xyz2 + ((size_t)bi * st.m + (live ? k : 0)) * 3;
    u4 ra[RECQ], rb[RECQ];
    point_records(q, meta + (size_t)bi * METAF, live, ra, rb);
    if (first) {
        u4 *o = const_cast<u4 *>(st.recB1) + ((size_t)bi * st.NP + k) * RECQ;
#pragma unroll
        for (int u = 0; u < RECQ; ++u) o[u] = rb[u];
        st.ratioL_p[(size_t)bi * st.NP + k] = 0.f;
    } else {
        const size_t at = (size_t)bi * st.MP + k;
#pragma unroll
        for (int u = 0; u < RECQ; ++u) { recA2_dense[at * RECQ + u] = ra[u]; st.recA2[0][at * RECQ + u] = ra[u]; }
        st.idx2[0][at] = k;
        st.remainR_c[0][at] = live ? multiR : 0.f;
        st.ratioR_c[at] = 0.f;
        if (live)
            for (int qv = 0; qv < NLEVEL; ++qv) ws[qv * lstride + (size_t)bi * st.rstride + st.n + k] = 0.f;
        if (k == 0) st.count[bi] = st.m;
    }
}

// The three passes of a level must see THE SAME weight for a pair: pass 1 divides remainL by sum_l w remainR, pass 3 takes
// sum_l w ratioL ratioR back off remainL -- for a point the level consumes entirely the two cancel, and what is left
// (rounding) is divided by the next level's, possibly tiny, sum.  With weights that agree to 1e-7 that residue is harmless
// (the packed-VALU kernels compute bit-identical d^2 in all passes); with weights that differ by the expanded form's 1e-3 at
// the steep levels it grew to O(0.3) changes of the matching (measured, r05: a first version ran pass 2 with the clouds'
// operand roles exchanged and pass 3 with (exp2 of the next level's exponent)^4).  Hence: cloud 2 is ALWAYS the A operand
// (rows) and cloud 1 ALWAYS the B operand (columns), every pass of level j scales them by the same (fa_j, fb_j), and every
// weight is exp2 of that MFMA's result -- the same bits in passes 1, 2, 3 and in the materialisation.

// fragments' accumulator register r of lane (half, col): row 8 (r / 4) + 4 half + r % 4, column col
__device__ __forceinline__ float pick4(const float4 (&w)[4], int r) {
    return r % 4 == 0 ? w[r / 4].x : r % 4 == 1 ? w[r / 4].y : r % 4 == 2 ? w[r / 4].z : w[r / 4].w;
}
__device__ __forceinline__ int pick4i(const int4 (&w)[4], int r) {
    return r % 4 == 0 ? w[r / 4].x : r % 4 == 1 ? w[r / 4].y : r % 4 == 2 ? w[r / 4].z : w[r / 4].w;
}
__device__ __forceinline__ h8 scale8(u4 v, u4 sv) {          // four exact v_pk_mul_f16 (powers of two)
    return __builtin_bit_cast(h8, v) * __builtin_bit_cast(h8, sv);
}
struct Frag { h8 q1, q2; };                                    // a lane's operands of the two MFMAs of a tile
__device__ __forceinline__ Frag scale_frag(u4 r1, u4 r2, const u4 (&sv)[2]) { return Frag{scale8(r1, sv[0]), scale8(r2, sv[1])}; }
// The MFMA of a tile: the builtin with a REGISTER zero as its C operand (`zacc`, sixteen VGPRs made opaque to the compiler).
// Why not the literal 0 -- r05, measured on MI355X (tests/diag/emd_repeat.py, emd_flake_rate.py): with C = 0 the compiler
// selects the instruction form whose destination may share registers with a dying source ("v_mfma_f32_32x32x16_f16 v[2:17],
// v[70:73], v[2:5], 0"), and in the builds where it did the sparse-regime pass 2 returned a few sums per million that differed
// from run to run.  With C in registers the compiler uses the early-clobber form (destination apart from all three sources) and
// keeps its own hazard bookkeeping.  Why not inline asm with an early-clobber output and written-out wait states (the first
// remedy tried): bit-stable at 2 x 2048^2, but at 8192^2 with two or more waves per SIMD pass 1 itself then returned ~1 (B = 2)
// to ~50 (B = 16) differing sums per call -- more with more s_nop, none with one wave per SIMD, none with the builtin.  The
// overlap alone is harmless (tools/ubench/mfma_overlap.hip: 2e7 isolated MFMAs per pattern).  r06 established the cause inside
// the scheduled code: every one of those builds was vectorised, the vectoriser's packed accumulations use the form
// "v_pk_fma_f32 ... op_sel:[0,1,0]", and that form loses its low half in lanes 48-63 while another wave of the SIMD issues MFMAs
// at certain distances (tools/ubench/pk_vs_mfma_forms.hip) -- "two or more waves per SIMD", "more with more s_nop" and "grew with
// the kernel's duration" are that condition.  Which forms repeat bit for bit: test_matrix_core_passes_repeat_bit_for_bit
// runs the sizes and regimes that flickered.  tools/mfma_overlap_check.py lists overlapping MFMAs in any .s
// (tests/test_isa_cpu.py: none in emd.hip; the flow / Chamfer / encoder kernels have some, all first-of-chain, and their
// bit-exact and replay-equals-eager tests have never flickered).
__device__ __forceinline__ f16acc opaque_zero() {
    f16acc z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(z));
    return z;
}
#ifndef EMD_DBG
#define EMD_DBG 0          // A/B builds of r06's hunt (tests/diag/emd_rows_probe.py, emd_chain_mismatch.py; all of them flickered or
                           // agreed exactly as the plain build did -- DESIGN 4.6): 1 64 wait states behind the chain, 2 no chaining,
                           // 4 tiles serialised, 8 chained against unchained in every call, mismatches recorded
#endif
#if EMD_DBG & 8
__device__ unsigned g_emd_dbg_n;
__device__ float g_emd_dbg[16 * 8];
#endif
__device__ __forceinline__ f16acc pair_exponents(const Frag &rows, const Frag &cols, const f16acc &zacc) {
#if EMD_DBG & 8
    {   // chained against unchained, mismatches recorded
        const f16acc i1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q1, cols.q1, zacc, 0, 0, 0);
        const f16acc ch = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q2, cols.q2, i1, 0, 0, 0);
        const f16acc i2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q1, cols.q1, zacc, 0, 0, 0);
        const f16acc rs = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q2, cols.q2, zacc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float alt = i2[r] + rs[r];
            if (!(fabsf(alt - ch[r]) <= 1e-2f + 1e-4f * fabsf(alt))) {
                const unsigned slot = atomicAdd(&g_emd_dbg_n, 1u);
                if (slot < 16) {
                    g_emd_dbg[slot * 8 + 0] = ch[r]; g_emd_dbg[slot * 8 + 1] = alt; g_emd_dbg[slot * 8 + 2] = i2[r]; g_emd_dbg[slot * 8 + 3] = rs[r];
                    g_emd_dbg[slot * 8 + 4] = (float)threadIdx.x; g_emd_dbg[slot * 8 + 5] = (float)r;
                    g_emd_dbg[slot * 8 + 6] = (float)blockIdx.x; g_emd_dbg[slot * 8 + 7] = (float)blockIdx.y;
                }
            }
        }
        return ch;
    }
#elif EMD_DBG & 2
    f16acc ints = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q1, cols.q1, zacc, 0, 0, 0);
    f16acc rest = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q2, cols.q2, zacc, 0, 0, 0);
    return ints + rest;
#else
    const f16acc ints = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q1, cols.q1, zacc, 0, 0, 0);
    f16acc acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rows.q2, cols.q2, ints, 0, 0, 0);
#if EMD_DBG & 1
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc));
#endif
#if EMD_DBG & 4
    __builtin_amdgcn_sched_barrier(0);
#endif
    return acc;
#endif
}

// EMD_PIN_LOADS=1 (an A/B build, off by default): keep every prefetch behind the VALU instructions that consumed the previous
// tile's MFMA results, and every tile's MFMAs behind the prefetch in front of it.  Built in r06 to test ONE hypothesis for r05's
// run-to-run differences -- the compiler counts a register dead once its last reader has issued and hands a scaled fragment's
// registers to the next prefetch ("v_mfma ... v[122:125] ...; global_load_dwordx4 v[122:125]" behind four queued MFMAs);
// tools/mfma_overlap_check.py --war lists such sites.  REJECTED: tools/ubench/mfma_war.hip and mfma_valu_war.hip overwrite the
// sources of up to seven queued MFMAs by a cache-resident load / by VALU writes in the very next slot, 1.6e7 times at up to four
// waves per SIMD, without one wrong result -- the hardware interlocks it; the pinned build flickered like the unpinned one, and
// both repeat once the vectoriser is off (what the vectoriser wrote was "v_pk_fma_f32 ... op_sel:[0,1,0]", a packed form
// that loses its low half in lanes 48-63 beside another wave's MFMAs -- tools/ubench/pk_vs_mfma_forms.hip).  The pins cost nothing
// measurable (cfg5 2.740 vs 2.741 ms).
#ifndef EMD_PIN_LOADS
#define EMD_PIN_LOADS 0
#endif
// ... and the MFMAs of the tile that follows a prefetch must not be moved IN FRONT of it (an MFMA is no memory operation: the
// scheduler hoists it over a plain memory barrier): the tile's raw fragments pass through the barrier
__device__ __forceinline__ void mfmas_stay_behind(u4 &r) {
#if EMD_PIN_LOADS
    asm volatile("" : "+v"(r) : : "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
template <int N>
__device__ __forceinline__ void loads_stay_behind(float (&v)[N]) {
#if EMD_PIN_LOADS
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]) : : "memory");
#endif
}

// Passes over cloud 1's points (columns; cloud 2's LIVE points stream past as row tiles with their weights):
// MODE 0: pass 1          s_k = sum_l w remainR[l];                     ratioL[k] = remainL[k] / (1e-9 + s_k)
// MODE 3: pass 3          s_k = sum_l w (remainR[l] - ratioR[l]);       remainL[k] = max(0, ratioL[k] (1e-9 + s_k))   [= remainL - ratioL sum w ratioR]
// lf: level j's 4^(j-7) as the per-slot factors of level_vectors.  rb_cur: the ratio slot of level j.
template <int MODE>
__global__ __launch_bounds__(64 * MSL) void emd_mfma_cols_kernel(MfmaState st, int cur, LevelFac lf, float *rb_cur) {
    __shared__ float part[MSL][MPW];
    if (gate_closed(st.gate, 0u)) return;
    const f16acc zacc = opaque_zero();
    const int bi = blockIdx.y, lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    const int half = lane >> 5, col = lane & 31;
    const u4 *ownrec = st.recB1 + ((size_t)bi * st.NP + blockIdx.x * MPW) * RECQ;
    const u4 *candrec = st.recA2[cur] + ((size_t)bi * st.MP + col) * RECQ + half;
    const float *w0 = (MODE == 0 ? st.remainR_c[cur] : st.ratioR_c) + (size_t)bi * st.MP + 4 * half;
    u4 svr[2], svc[2];
    level_vectors(lf, half, true, svr);
    level_vectors(lf, half, false, svc);
    Frag bf[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) bf[t] = scale_frag(ownrec[(t * 32 + col) * RECQ + half], ownrec[(t * 32 + col) * RECQ + 2 + half], svc);
    const int tiles = (st.count[bi] + 31) / 32;
    const int tb = (int)((long)tiles * slice / S), te = (int)((long)tiles * (slice + 1) / S);
    float sa[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[t] = 0.f;
    if (tb < te) {
        // two register sets A / B, each loaded one tile before its use and first touched after the other tile's arithmetic
        // (a rotating copy made the compiler wait for the load it had just issued -- a full memory round trip per tile
        // whenever few waves share the SIMD)
        constexpr size_t TQ = 32 * RECQ;                         // quarters per tile of records
        u4 afA[2] = {candrec[(size_t)tb * TQ], candrec[(size_t)tb * TQ + 2]}, afB[2];
        float4 waA[4], waB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) waA[i] = *(const float4 *)(w0 + tb * 32 + 8 * i);
        auto tile = [&](const u4 (&af)[2], const float4 (&wa)[4]) {
            const Frag as = scale_frag(af[0], af[1], svr);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const f16acc acc = pair_exponents(as, bf[t], zacc);
#pragma unroll
                for (int r = 0; r < 16; ++r) sa[t] = __builtin_fmaf(fast_exp2(acc[r]), pick4(wa, r), sa[t]);
            }
        };
        for (int ct = tb; ct < te; ct += 2) {
            const int n1 = min(ct + 1, te - 1), n2 = min(ct + 2, te - 1);
            afB[0] = candrec[(size_t)n1 * TQ]; afB[1] = candrec[(size_t)n1 * TQ + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) waB[i] = *(const float4 *)(w0 + n1 * 32 + 8 * i);
            mfmas_stay_behind(afA[0]); mfmas_stay_behind(afA[1]);
            tile(afA, waA);
            loads_stay_behind(sa);
            afA[0] = candrec[(size_t)n2 * TQ]; afA[1] = candrec[(size_t)n2 * TQ + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) waA[i] = *(const float4 *)(w0 + n2 * 32 + 8 * i);
            mfmas_stay_behind(afB[0]); mfmas_stay_behind(afB[1]);
            if (ct + 1 < te) tile(afB, waB);
            loads_stay_behind(sa);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        sa[t] += __shfl_xor(sa[t], 32);
        if (half == 0) part[slice][t * 32 + col] = sa[t];
    }
    __syncthreads();
    float *remainL = st.temp + (size_t)bi * (st.n + st.m) * 2;
    float *ratioL = rb_cur + (size_t)bi * st.rstride;
    for (int tid = slice * 64 + lane; tid < MPW; tid += 64 * S) {
        const int k = blockIdx.x * MPW + tid;
        if (k >= st.n) continue;
        float tot = 0.f;
        for (int u = 0; u < S; ++u) tot += part[u][tid];
        if (MODE == 0) {
            const float r = remainL[k] / (1e-9f + tot);
            ratioL[k] = r;
            st.ratioL_p[(size_t)bi * st.NP + k] = r;
        } else {
            // remainL - ratioL sum_l w ratioR  with  ratioL = remainL / (1e-9 + s),  s = pass 1's sum_l w remainR
            //   = ratioL (1e-9 + sum_l w (remainR[l] - ratioR[l])).
            // r06: the second form -- pass 3's row weights are the DEFICITS remainR - ratioR that pass 2 leaves (an exact fp32
            // subtraction of two numbers within a factor of two; 0 for every row the level consumes), so the sum has no negative
            // term and what is left of a consumed point is remainL 1e-9 / (1e-9 + s) as in exact arithmetic.  The first form
            // cancels two sums of size s and leaves +-1e-7 remainL of rounding noise, which the next level divides by ITS
            // (possibly 1e-9-sized) sum: measured on 32 x 32 and 33 x 33 cases of tests/diag/emd_matrix_fuzz.py -- ratioL of a
            // consumed point 45-120 x off, cost 2-5e-5 off the oracle on exponents that a float64 auction turns into the
            // oracle's cost to 4e-9 (tests/diag/emd_case_levels.py, emd_fuzz_case_vs_oracle.py).  The reference's loop
            // (approxmatch.cu:130-163) has that noise with its own realisation (the fp32 oracle sits 1e-8 .. 4e-6 from the
            // float64 auction); the packed-VALU family reproduces the reference's operations bit for bit.
            remainL[k] = fmaxf(0.0f, ratioL[k] * (1e-9f + tot));
        }
    }
}

// sum of v over the 32 lanes of the lane's half (every lane gets it)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

// Pass 2, over cloud 2's live points -- which stay the ROWS: a wave owns MT row tiles (their A fragments), cloud 1's points
// stream past as column tiles with ratioL (one weight per lane), the sums run along the accumulators' columns, i.e. across
// lanes: per-lane partial sums for every (tile, register) and one butterfly per such sum at the end.
// RT row tiles per wave: 4 while more than half of cloud 2 is live (regime 1); 1 below that (regime 2: a workgroup's waves then
// walk 32 x RT x n / S pairs in a row with nobody else on their SIMD -- with RT = 4 a sparse level took the same 100 us as the
// dense one).  Both are launched from the second level on; the one whose regime it is not returns at once.
//   sumr = remainR[l] * sum_k w ratioL[k];  ratioR[l] = min(remainR[l] / (sumr + 1e-9), 1) * remainR[l];
//   remainR[l] = max(0, remainR[l] - sumr)
template <int RT>
__global__ __launch_bounds__(64 * MSL) void emd_mfma_rows_kernel(MfmaState st, int cur, LevelFac lf, float *rb_cur, int regime) {
    constexpr int RPW = 32 * RT;                                    // rows per workgroup
    __shared__ float part[MSL][RPW];
    if (gate_closed(st.gate, 0u)) return;
    const f16acc zacc = opaque_zero();
    const int bi = blockIdx.y, lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    const int cnt = st.count[bi];
    const int rowblock = blockIdx.x;
    if (rowblock * RPW >= cnt) return;                              // (whole workgroup)
    if (regime == 1 ? 2 * cnt <= st.m : regime == 2 ? 2 * cnt > st.m : false) return;
    const int half = lane >> 5, col = lane & 31;
    const u4 *ownrec = st.recA2[cur] + ((size_t)bi * st.MP + rowblock * RPW) * RECQ;
    const u4 *candrec = st.recB1 + ((size_t)bi * st.NP + col) * RECQ + half;
    const float *wl = st.ratioL_p + (size_t)bi * st.NP;
    u4 svr[2], svc[2];
    level_vectors(lf, half, true, svr);
    level_vectors(lf, half, false, svc);
    Frag af[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) af[t] = scale_frag(ownrec[(t * 32 + col) * RECQ + half], ownrec[(t * 32 + col) * RECQ + 2 + half], svr);
    const int tiles = round_up(st.n, 32) / 32;
    const int tb = (int)((long)tiles * slice / S), te = (int)((long)tiles * (slice + 1) / S);
    float s[RT][16];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    if (tb < te) {
        // register sets A / B as in emd_mfma_cols_kernel, NB column tiles each: in the sparse regime a wave is alone on its SIMD
        // and a tile's arithmetic (~400 cycles) is far shorter than the loads' round trip -- four tiles per set there
        constexpr int NB = RT == 1 ? 4 : 1;
        constexpr size_t TQ = 32 * RECQ;
        u4 bfA[NB][2], bfB[NB][2];
        float wA[NB], wB[NB];
        auto load = [&](int c0, u4 (&bfr)[NB][2], float (&w)[NB]) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int c = min(c0 + u, te - 1);
                bfr[u][0] = candrec[(size_t)c * TQ]; bfr[u][1] = candrec[(size_t)c * TQ + 2];
                w[u] = c0 + u < te ? wl[c * 32 + col] : 0.f;          // (a repeated last tile weighs nothing)
            }
        };
        auto tiles_of = [&](const u4 (&bfr)[NB][2], const float (&w)[NB]) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const Frag bs = scale_frag(bfr[u][0], bfr[u][1], svc);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const f16acc acc = pair_exponents(af[t], bs, zacc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] = __builtin_fmaf(fast_exp2(acc[r]), w[u], s[t][r]);
                }
            }
        };
        auto fence = [&](u4 (&bfr)[NB][2]) {
#pragma unroll
            for (int u = 0; u < NB; ++u) { mfmas_stay_behind(bfr[u][0]); mfmas_stay_behind(bfr[u][1]); }
        };
        auto pin = [&]() {                       // (one sum per row tile: each depends on that tile's last MFMA)
            float last[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) last[t] = s[t][15];
            loads_stay_behind(last);
#pragma unroll
            for (int t = 0; t < RT; ++t) s[t][15] = last[t];
        };
        load(tb, bfA, wA);
        for (int ct = tb; ct < te; ct += 2 * NB) {
            load(ct + NB, bfB, wB);
            fence(bfA);
            tiles_of(bfA, wA);
            pin();
            load(ct + 2 * NB, bfA, wA);
            fence(bfB);
            if (ct + NB < te) tiles_of(bfB, wB);
            pin();
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float tot = half_wave_sum(s[t][r]);
            if (col == r) part[slice][t * 32 + 8 * (r / 4) + 4 * half + r % 4] = tot;
        }
    __syncthreads();
    float *remainR = st.temp + (size_t)bi * (st.n + st.m) * 2 + st.n;
    float *ratioR = rb_cur + (size_t)bi * st.rstride + st.n;
    for (int tid = slice * 64 + lane; tid < RPW; tid += 64 * S) {
        const int pos = rowblock * RPW + tid;
        if (pos >= cnt) continue;
        const size_t at = (size_t)bi * st.MP + pos;
        const int l = st.idx2[cur][at];
        float tot = 0.f;
        for (int u = 0; u < S; ++u) tot += part[u][tid];
        const float rr = st.remainR_c[cur][at];
        const float sumr = tot * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        const float r = consumption * rr, rem = fmaxf(0.0f, rr - sumr);
        ratioR[l] = r;
        st.ratioR_c[at] = rr - r;                  // pass 3's weight: the deficit (cols kernel, MODE 3)
        remainR[l] = rem;
        st.remainR_c[cur][at] = rem;
    }
}

// After a level's three passes: the points whose remainR is still nonzero move to the other buffer, in order; the others take
// the places [kept, cnt) of `order` (LAST: every point of the last list takes [0, cnt)).  One workgroup per cloud.
template <bool LAST>
__global__ __launch_bounds__(1024) void emd_mfma_compact_kernel(MfmaState st, int cur, int q) {
    __shared__ int wsum[2][16];
    __shared__ int total;
    if (gate_closed(st.gate, 0u)) return;
    const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nxt = cur ^ 1;
    const int cnt = st.count[bi];
    const size_t base = (size_t)bi * st.MP;
    if (tid == 0) { st.counts[q * st.nb + bi] = cnt; total = 0; }
    __syncthreads();
    int kept_total = 0;
    if (!LAST) {
        int c = 0;
        for (int pos = tid; pos < cnt; pos += 1024) c += st.remainR_c[cur][base + pos] != 0.f;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) atomicAdd(&total, c);                       // (integer: order-independent)
        __syncthreads();
        kept_total = total;
    }
    int kbase = 0, dbase = kept_total;
    for (int c0 = 0; c0 < cnt; c0 += 1024) {
        const int pos = c0 + tid;
        const bool valid = pos < cnt;
        const bool keep = !LAST && valid && st.remainR_c[cur][base + (valid ? pos : 0)] != 0.f;
        const bool drop = valid && !keep;
        const unsigned long long bk = __ballot(keep), bd = __ballot(drop);
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        const int pk = __popcll(bk & below), pd = __popcll(bd & below);
        if (lane == 0) { wsum[0][wave] = __popcll(bk); wsum[1][wave] = __popcll(bd); }
        __syncthreads();
        int wk = 0, wd = 0, ck = 0, cd = 0;
        for (int u = 0; u < 16; ++u) {
            if (u < wave) { wk += wsum[0][u]; wd += wsum[1][u]; }
            ck += wsum[0][u]; cd += wsum[1][u];
        }
        if (keep) {
            const size_t from = base + pos, to = base + kbase + wk + pk;
#pragma unroll
            for (int u = 0; u < RECQ; ++u) st.recA2[nxt][to * RECQ + u] = st.recA2[cur][from * RECQ + u];
            st.idx2[nxt][to] = st.idx2[cur][from];
            st.remainR_c[nxt][to] = st.remainR_c[cur][from];
        } else if (drop) {
            st.order[base + dbase + wd + pd] = st.idx2[cur][base + pos];
        }
        kbase += ck; dbase += cd;
        __syncthreads();
    }
    if (!LAST) {
        // the next list's last tile: zero weights behind its end, records that are numbers
        const int pad_end = min(round_up(kept_total, 32), st.MP);
        for (int pos = kept_total + tid; pos < pad_end; pos += 1024) {
            const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < RECQ; ++u) st.recA2[nxt][(base + pos) * RECQ + u] = z;
            st.remainR_c[nxt][base + pos] = 0.f;
        }
        for (int pos = kept_total + tid; pos < min(round_up(cnt, 32), st.MP); pos += 1024) st.ratioR_c[base + pos] = 0.f;
        if (tid == 0) st.count[bi] = kept_total;
    }
}

// Cloud 2 in `order`: A records, every level's ratioR, coordinates (by planes, for the cost) and original row numbers (-1 in
// the padding)
__global__ void emd_mfma_gather_kernel(MfmaState st, const u4 *__restrict__ recA2_dense, const float *__restrict__ xyz2,
                                       const float *__restrict__ ws, size_t lstride, u4 *__restrict__ recA_s,
                                       float *__restrict__ rr_s, float *__restrict__ c2soa_s, int *__restrict__ l_s) {
    if (gate_closed(st.gate, 0u)) return;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
    if (pos >= st.MP) return;
    const bool valid = pos < st.m;
    const size_t at = (size_t)bi * st.MP + pos;
    const int l = valid ? st.order[at] : 0;
    const u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < RECQ; ++u) recA_s[at * RECQ + u] = valid ? recA2_dense[((size_t)bi * st.MP + l) * RECQ + u] : z;
    for (int qv = 0; qv < NLEVEL; ++qv)
        rr_s[((size_t)qv * st.nb + bi) * st.MP + pos] = valid ? ws[qv * lstride + (size_t)bi * st.rstride + st.n + l] : 0.f;
    const float *p = xyz2 + ((size_t)bi * st.m + l) * 3;
    for (int u = 0; u < 3; ++u) c2soa_s[((size_t)bi * 3 + u) * st.MP + pos] = valid ? p[u] : 0.f;
    l_s[at] = valid ? l : -1;
}

struct LevelScales { LevelFac lf[NLEVEL]; };

// match[l][k] = sum over the levels, in level order, of w_j(l, k) ratioL_j[k] ratioR_j[l] with the SAME w_j the passes saw (the
// same MFMA on the same operands) -- with the difference form's weights here the rows and columns of `match` summed to 1 +- 4e-3
// (the expanded form's error at the steep levels) instead of 1 +- 1e-6.  A wave owns MTM 32-column tiles of cloud 1 and takes
// every S-th 32-row tile of cloud 2 IN `order`: tile T gets the levels whose list is longer than 32 T (the others' ratioR are
// all zero there) -- per (pair, live level): exp2, a multiplication, an FMA.  The accumulator layout writes 128 contiguous
// bytes of a `match` row per register and half-wave.
// COST: sum match * |x1 - x2| with the distance in the difference form, partial sums per (cloud, workgroup, slice) for
// emd_cost_sum_kernel -- fixed order, deterministic.
// Measured (cfg5, 16 x 8192^2): 1.18 ms = `match` written at 3.6 TB/s; MTM = 4 (512 contiguous bytes per row and wave) and the
// unmasked store path moved it by < 10 %: it is the write that bounds it, not the instruction count.
constexpr int MTM = 2;
#ifndef EMD_NT_STORE
#define EMD_NT_STORE 1
#endif
#if EMD_NT_STORE
// `match` (4 n m bytes: 4.3 GB at cfg5) is written once and not read again by the call: a streaming store.  r06, measured at cfg5
// (B = 16, N = 8192): emd_mfma_materialize_kernel 1.31 -> 0.82 ms, the approxmatch + cost call 2.73 -> 2.24 ms
#define EMD_MATCH_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define EMD_MATCH_STORE(v, p) (*(p) = (v))
#endif
template <bool COST>
// (no __restrict__ on what the loop loads: a load from memory the compiler knows nobody writes may cross the asm barriers that
// keep loads and MFMAs apart -- loads_stay_behind / mfmas_stay_behind -- and r06's first build sank the last ratioR load of a
// level behind that level's MFMAs, into the registers of a fragment they read)
__global__ __launch_bounds__(64 * MSL) void emd_mfma_materialize_kernel(MfmaState st, LevelScales ls, const float *xyz1,
                                                                        const float *ws, size_t lstride,
                                                                        const u4 *recA_s, const float *rr_s,
                                                                        const float *c2soa_s, const int *l_s,
                                                                        float *match, float *costpart) {
    __shared__ float rl_s[MSL][NLEVEL][32 * MTM];     // ratioL of the wave's columns, by level (a register array would be indexed
    __shared__ u4 sv_s[NLEVEL][2][2][2];              //   dynamically by the level loop -- scratch -- or unrolled nine times -- spills)
                                                      // sv_s[level][rows | columns][half][MFMA]: level_vectors
    __shared__ int cnt_s[NLEVEL];
    if (gate_closed(st.gate, 0u)) return;
    const f16acc zacc = opaque_zero();
    const int bi = blockIdx.y, nb = gridDim.y, lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    const int half = lane >> 5, col = lane & 31;
    const u4 *ownrec = st.recB1 + ((size_t)bi * st.NP + blockIdx.x * (32 * MTM)) * RECQ;
    const u4 *candrec = recA_s + ((size_t)bi * st.MP + col) * RECQ + half;
    float *mt = match + (size_t)bi * st.n * st.m;
    u4 bfraw[MTM][2];
    float px[MTM], py[MTM], pz[MTM];
    int kk[MTM];
    if (slice == 0 && lane < 4 * NLEVEL) {             // (level, side, half) -> its two vectors
        u4 v[2];
        level_vectors(ls.lf[lane >> 2], lane & 1, (lane & 2) == 0, v);
        sv_s[lane >> 2][(lane >> 1) & 1][lane & 1][0] = v[0];
        sv_s[lane >> 2][(lane >> 1) & 1][lane & 1][1] = v[1];
    }
    if (slice == 0 && lane < NLEVEL) cnt_s[lane] = st.counts[lane * nb + bi];
#pragma unroll
    for (int t = 0; t < MTM; ++t) {
        bfraw[t][0] = ownrec[(t * 32 + col) * RECQ + half];
        bfraw[t][1] = ownrec[(t * 32 + col) * RECQ + 2 + half];
        kk[t] = blockIdx.x * (32 * MTM) + t * 32 + col;
        const int kc = min(kk[t], st.n - 1);
        if (half == 0)
#pragma unroll
            for (int j = 0; j < NLEVEL; ++j) rl_s[slice][j][t * 32 + col] = ws[j * lstride + (size_t)bi * st.rstride + kc];
        if (COST) {
            const float *p = xyz1 + ((size_t)bi * st.n + kc) * 3;
            px[t] = p[0]; py[t] = p[1]; pz[t] = p[2];
        }
    }
    __syncthreads();
    const int tiles = round_up(st.m, 32) / 32;
    float cost = 0.f;
    for (int ct = slice; ct < tiles; ct += S) {
        const u4 af[2] = {candrec[(size_t)ct * (32 * RECQ)], candrec[(size_t)ct * (32 * RECQ) + 2]};
        f16acc mm[MTM];
#pragma unroll
        for (int t = 0; t < MTM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mm[t][r] = 0.f;
#pragma unroll 1
        for (int j = 0; j < NLEVEL && cnt_s[j] > ct * 32; ++j) {
            float4 rr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rr[i] = *(const float4 *)(rr_s + ((size_t)j * nb + bi) * st.MP + ct * 32 + 8 * i + 4 * half);
            const u4 svr[2] = {sv_s[j][0][half][0], sv_s[j][0][half][1]}, svc[2] = {sv_s[j][1][half][0], sv_s[j][1][half][1]};
            float rlj[MTM];
#pragma unroll
            for (int t = 0; t < MTM; ++t) rlj[t] = rl_s[slice][j][t * 32 + col];
            // every load of this level is issued in front of its first MFMA (the raw row fragment passes through the barrier
            // the MFMAs' operands are made behind), none before its last MFMA has been consumed (below): mfmas_stay_behind
            u4 afj[2] = {af[0], af[1]};
            mfmas_stay_behind(afj[0]); mfmas_stay_behind(afj[1]);
            const Frag as = scale_frag(afj[0], afj[1], svr);
#pragma unroll
            for (int t = 0; t < MTM; ++t) {
                const f16acc acc = pair_exponents(as, scale_frag(bfraw[t][0], bfraw[t][1], svc), zacc);
#pragma unroll
                for (int r = 0; r < 16; ++r) mm[t][r] = __builtin_fmaf(fmul(fast_exp2(acc[r]), rlj[t]), pick4(rr, r), mm[t][r]);
            }
            {   // the next level's loads stay behind this level's MFMAs (loads_stay_behind)
                float last[MTM];
#pragma unroll
                for (int t = 0; t < MTM; ++t) last[t] = mm[t][15];
                loads_stay_behind(last);
#pragma unroll
                for (int t = 0; t < MTM; ++t) mm[t][15] = last[t];
            }
        }
        // register 4 i + jj of lane (half, col): the row at place ct 32 + 8 i + 4 half + jj of `order`, columns kk[0] and kk[0] + 32.
        // FULL (wave-uniform, all but the last tiles): no lane masks anywhere -- the masked form of this stage was 1 200
        // instructions per row tile, twice the level loop
        auto emit = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 l4 = *(const int4 *)(l_s + (size_t)bi * st.MP + ct * 32 + 8 * i + 4 * half);
                float4 qx, qy, qz;
                if (COST) {
                    const float *c = c2soa_s + (size_t)bi * 3 * st.MP + ct * 32 + 8 * i + 4 * half;
                    qx = *(const float4 *)c; qy = *(const float4 *)(c + st.MP); qz = *(const float4 *)(c + 2 * (size_t)st.MP);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int l = jj == 0 ? l4.x : jj == 1 ? l4.y : jj == 2 ? l4.z : l4.w;
                    float *row = mt + (size_t)(FULL ? l : max(l, 0)) * st.n + kk[0];
                    const float cx = jj == 0 ? qx.x : jj == 1 ? qx.y : jj == 2 ? qx.z : qx.w;
                    const float cy = jj == 0 ? qy.x : jj == 1 ? qy.y : jj == 2 ? qy.z : qy.w;
                    const float cz = jj == 0 ? qz.x : jj == 1 ? qz.y : jj == 2 ? qz.z : qz.w;
#pragma unroll
                    for (int t = 0; t < MTM; ++t) {
                        const float v = mm[t][4 * i + jj];
                        const bool live = FULL || (l >= 0 && kk[t] < st.n);
                        if (live) EMD_MATCH_STORE(v, &row[32 * t]);
                        // the distance only where some entry of the wave's 2 x 32 is worth it: almost every pair's weight is below
                        // 1e-12 of a matched pair's (what is skipped sums to < 1e-6 of the cost)
                        if (COST && __ballot(live && v > 1e-12f) != 0ull) {
                            const float d = __builtin_amdgcn_sqrtf(sqdist(px[t], py[t], pz[t], cx, cy, cz));
                            cost = __builtin_fmaf(live ? v : 0.f, d, cost);
                        }
                    }
                }
            }
        };
        if (ct * 32 + 32 <= st.m && (int)blockIdx.x * (32 * MTM) + 32 * MTM <= st.n) emit(std::true_type{});
        else emit(std::false_type{});
    }
    if (COST) {
        for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
        if (lane == 0) costpart[((size_t)bi * gridDim.x + blockIdx.x) * S + slice] = cost;
    }
}

// Debug / test: the exp2 arguments of every pair of one cloud pair at one level, exactly as the passes form them (same records,
// same level vectors, same two MFMAs): out[l * n + k], l a point of cloud 2, k of cloud 1.  One 32 x 32 tile per wave.
__global__ __launch_bounds__(64) void emd_mfma_debug_exponents_kernel(MfmaState st, LevelFac lf, const u4 *__restrict__ recA2_dense,
                                                                      float *__restrict__ out, unsigned slot_mask) {
    if (gate_closed(st.gate, 0u)) return;
    const f16acc zacc = opaque_zero();
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    u4 svr[2], svc[2];
    level_vectors(lf, half, true, svr);
    level_vectors(lf, half, false, svc);
    const u4 *rb = st.recB1 + ((size_t)blockIdx.x * 32 + col) * RECQ, *ra = recA2_dense + ((size_t)blockIdx.y * 32 + col) * RECQ;
    Frag fr = scale_frag(ra[half], ra[2 + half], svr);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                   // (slot_mask: bit k = K slot k of MFMA 1, bit 16 + k of MFMA 2)
        if (!((slot_mask >> (8 * half + i)) & 1u)) fr.q1[i] = (_Float16)0.f;
        if (!((slot_mask >> (16 + 8 * half + i)) & 1u)) fr.q2[i] = (_Float16)0.f;
    }
    const f16acc acc = pair_exponents(fr, scale_frag(rb[half], rb[2 + half], svc), zacc);
    const int k = blockIdx.x * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = blockIdx.y * 32 + 8 * (r / 4) + 4 * half + r % 4;
        if (l < st.m && k < st.n) out[(size_t)l * st.n + k] = acc[r];
    }
}

int pick_mfma_slices(int b, int npoints, int ninner) {
    const long groups = (long)b * ((npoints + MPW - 1) / MPW);
    const int tiles = (ninner + 31) / 32;
    int s = 1;
    while (s < MSL && groups * s < 4096 && tiles / (2 * s) >= 4) s *= 2;
    return s;
}

bool g_matrix_path = true;


// (xyz2_l, ratioR_level0..8[l]) records of 12 floats for the materialisation
__global__ void emd_pack_levels_kernel(int n, int m, const float *__restrict__ xyz2, const float *__restrict__ ws,
                                       size_t lstride, size_t rstride, float *__restrict__ rec, const unsigned *gate) {
    if (gate_closed(gate, 1u)) return;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= m) return;
    const int bi = blockIdx.y;
    const float *q = xyz2 + ((size_t)bi * m + l) * 3;
    float4 *o = (float4 *)(rec + ((size_t)bi * m + l) * 12);
    float w[NLEVEL];
#pragma unroll
    for (int j = 0; j < NLEVEL; ++j) w[j] = ws[j * lstride + (size_t)bi * rstride + n + l];
    o[0] = make_float4(q[0], q[1], q[2], w[0]);
    o[1] = make_float4(w[1], w[2], w[3], w[4]);
    o[2] = make_float4(w[5], w[6], w[7], w[8]);
}

struct LevelPairs { u64 p[(NLEVEL + 1) / 2]; };   // lvl2[2i] | lvl2[2i+1] << 32

template <int J>
__device__ __forceinline__ f2 level_term(const LevelPairs &lv, const u64 *r, const f2 (&rl)[NLEVEL], f2 d2) {
    // record floats: 0..2 xyz, 3+J the level's weight -> pair (3+J)>>1, half (3+J)&1
    const f2 e = exp2_2(bmul<J & 1>(lv.p[J >> 1], d2));
    return bmul<(3 + J) & 1>(r[(3 + J) >> 1], e * rl[J]);
}
template <int J>
__device__ __forceinline__ f2 level_sum(const LevelPairs &lv, const u64 *r, const f2 (&rl)[NLEVEL], f2 d2) {
    if constexpr (J == 0) return level_term<0>(lv, r, rl, d2);
    else return level_sum<J - 1>(lv, r, rl, d2) + level_term<J>(lv, r, rl, d2);
}

// COST: also the matching's cost sum_{l,k} match[l][k] * |x1_k - x2_l| (approxmatch.cu:184-224) of this wave's
// pairs -- the distance is already there -- to costpart[(cloud, workgroup, slice)]; emd_cost_sum_kernel adds a cloud's
// partials in a fixed order.  Saves matchcost's pass over `match` (4*n*m bytes per cloud).
template <bool COST>
__global__ __launch_bounds__(1024) void emd_materialize2_kernel(int n, int m, LevelPairs lv, const float *__restrict__ xyz1,
                                                                const float *__restrict__ rec, float *__restrict__ match,
                                                                const float *__restrict__ ws, size_t lstride, size_t rstride,
                                                                float *__restrict__ costpart, const unsigned *gate) {
    if (gate_closed(gate, 1u)) return;
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    float *__restrict__ mt = match + (size_t)bi * n * m;
    const int k0 = blockIdx.x * PPW + lane, k1 = k0 + 64;
    const int a0 = min(k0, n - 1), a1 = min(k1, n - 1);
    const f2 px = {P[a0 * 3 + 0], P[a1 * 3 + 0]}, py = {P[a0 * 3 + 1], P[a1 * 3 + 1]}, pz = {P[a0 * 3 + 2], P[a1 * 3 + 2]};
    f2 rl[NLEVEL];
#pragma unroll
    for (int j = 0; j < NLEVEL; ++j) {
        const float *r = ws + j * lstride + (size_t)bi * rstride;
        rl[j] = f2{r[a0], r[a1]};
    }
    const int lb = (int)((long)m * slice / S), le = (int)((long)m * (slice + 1) / S);
    f2 cost = {0.f, 0.f};
    stream_records<6, 2>((const u64 *)(rec + (size_t)bi * m * 12), lb, le, [&](int l, const u64 *r) {
        const f2 d2 = sqdist2(px, py, pz, r[0], r[1]);
        const f2 acc = level_sum<NLEVEL - 1>(lv, r, rl, d2);
        float *row = mt + (size_t)l * n;
        if (k0 < n) EMD_MATCH_STORE(acc.x, &row[k0]);
        if (k1 < n) EMD_MATCH_STORE(acc.y, &row[k1]);
        if (COST) cost = fma2(acc, f2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)}, cost);
    });
    if (COST) {
        float c = (k0 < n ? cost.x : 0.f) + (k1 < n ? cost.y : 0.f);
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) costpart[((size_t)bi * gridDim.x + blockIdx.x) * S + slice] = c;
    }
}

// out[b] = the cloud's partial costs, added in a fixed order
__global__ __launch_bounds__(256) void emd_cost_sum_kernel(int nper, const float *__restrict__ costpart, float *__restrict__ out,
                                                           const unsigned *gate, unsigned want) {
    __shared__ float red[4];
    if (gate != nullptr && __builtin_nontemporal_load(gate) != want) return;
    const float *cp = costpart + (size_t)blockIdx.x * nper;
    float c = 0.f;
    for (int i = threadIdx.x; i < nper; i += 256) c += cp[i];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b] = sum_{l,k} match[b,l,k] * |xyz1[k] - xyz2[l]|                           approxmatch.cu:184-224
__global__ __launch_bounds__(256) void emd_cost_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2,
                                                       const float *__restrict__ match, float *out) {
    __shared__ float red[4];
    const int bi = blockIdx.y;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + (size_t)bi * m * 3;
    const float *__restrict__ mt = match + (size_t)bi * n * m;
    const int lb = (int)((long)m * blockIdx.x / gridDim.x), le = (int)((long)m * (blockIdx.x + 1) / gridDim.x);
    float sub = 0.f;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const float px = P[k * 3 + 0], py = P[k * 3 + 1], pz = P[k * 3 + 2];
        for (int l = lb; l < le; ++l) {
            const float d = sqrtf(sqdist(px, py, pz, Q[l * 3 + 0], Q[l * 3 + 1], Q[l * 3 + 2]));
            sub += mt[(size_t)l * n + k] * d;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sub += __shfl_xor(sub, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sub;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + bi, (red[0] + red[1]) + (red[2] + red[3]));
}

// grad1[k] = sum_l match[l][k] * (x1_k - x2_l) / max(|x1_k - x2_l|, 1e-10)         approxmatch.cu:270-291
__global__ __launch_bounds__(1024) void emd_grad1_kernel(int n, int m, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2,
                                                         const float *__restrict__ match, float *__restrict__ grad1) {
    __shared__ float part[MAXS][3][64];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x, slice = __builtin_amdgcn_readfirstlane(threadIdx.y), S = blockDim.y;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + (size_t)bi * m * 3;
    const float *__restrict__ mt = match + (size_t)bi * n * m;
    const int k = blockIdx.x * 64 + lane;
    const bool live = k < n;
    const int kc = min(k, n - 1);
    const float px = P[kc * 3 + 0], py = P[kc * 3 + 1], pz = P[kc * 3 + 2];
    const int lb = (int)((long)m * slice / S), le = (int)((long)m * (slice + 1) / S);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    stream_candidates_scalar(Q, lb, le, [&](int l, float qx, float qy, float qz) {
        const float dx = px - qx, dy = py - qy, dz = pz - qz;
        const float d = mt[(size_t)l * n + kc] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
        gx += dx * d; gy += dy * d; gz += dz * d;
    });
    part[slice][0][lane] = gx; part[slice][1][lane] = gy; part[slice][2][lane] = gz;
    __syncthreads();
    if (slice != 0 || !live) return;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    for (int u = 0; u < S; ++u) { tx += part[u][0][lane]; ty += part[u][1][lane]; tz += part[u][2][lane]; }
    float *g = grad1 + ((size_t)bi * n + k) * 3;
    g[0] = tx; g[1] = ty; g[2] = tz;
}

// grad2[l] = sum_k match[l][k] * (x2_l - x1_k) / max(|x2_l - x1_k|, 1e-10)         approxmatch.cu:229-269
// one wave per row l (coalesced over k), 4 rows per workgroup
__global__ __launch_bounds__(256) void emd_grad2_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match, float *__restrict__ grad2) {
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= m) return;
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + (size_t)bi * m * 3;
    const float *__restrict__ row = match + (size_t)bi * n * m + (size_t)l * n;
    const float qx = Q[l * 3 + 0], qy = Q[l * 3 + 1], qz = Q[l * 3 + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int k = lane; k < n; k += 64) {
        const float dx = qx - P[k * 3 + 0], dy = qy - P[k * 3 + 1], dz = qz - P[k * 3 + 2];
        const float d = row[k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
        gx += dx * d; gy += dy * d; gz += dz * d;
    }
    for (int o = 32; o > 0; o >>= 1) {
        gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); gz += __shfl_xor(gz, o);
    }
    if (lane == 0) {
        float *g = grad2 + ((size_t)bi * m + l) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
}

// Both gradients in ONE pass over `match` (the two kernels above read it twice: 8.6 GB at B=16, N=8192).
// A workgroup owns 256 columns k; every thread walks all m rows for its k, 21 rows at a time:
//   grad1[k] accumulates in registers;
//   grad2[l] = -sum_k (x1_k - x2_l) * c  needs a sum over the lanes for every row: the 63 values of a row
//   group go through one recursive-halving butterfly (63 shuffles instead of 378), the four waves combine in
//   LDS, and the workgroup's partial for those rows goes to part2[b][k-block][l][3]; emd_grad2_sum_kernel adds
//   the k-blocks in a fixed order.
constexpr int GROWS = 21;
template <int K>
__device__ __forceinline__ void halve_stage(float (&w)[32], int lane) {
    constexpr int n2 = 16 >> (K - 1);
    const bool up = (lane >> K) & 1;
#pragma unroll
    for (int i = 0; i < n2; ++i) {
        const float keep = up ? w[i + n2] : w[i];
        const float send = up ? w[i] : w[i + n2];
        w[i] = keep + __shfl_xor(send, 1 << K);
    }
}
// sum over the 64 lanes of 64 per-lane values t[0..63]; lane j returns the total of value bitreverse6(j)
__device__ __forceinline__ float lane_sums64(const float (&t)[64], int lane) {
    float w[32];
    const bool up = lane & 1;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float keep = up ? t[i + 32] : t[i];
        const float send = up ? t[i] : t[i + 32];
        w[i] = keep + __shfl_xor(send, 1);
    }
    halve_stage<1>(w, lane); halve_stage<2>(w, lane); halve_stage<3>(w, lane); halve_stage<4>(w, lane); halve_stage<5>(w, lane);
    return w[0];
}

// RS row slices per workgroup (RS * 256 threads): slice s walks the rows [s*mh, (s+1)*mh) -- the launch has only
// B * n / 64 column-waves (two per SIMD at B=16, n=8192), RS = 2 doubles the waves that hide each other's latency.
template <int RS>
__global__ __launch_bounds__(256 * RS) void emd_grad_fused_kernel(int n, int m, const float *__restrict__ xyz1,
                                                                  const float *__restrict__ xyz2,
                                                                  const float *__restrict__ match, float *__restrict__ grad1,
                                                                  float *__restrict__ part2) {
    __shared__ float red[RS][4][64];
    __shared__ float g1s[RS > 1 ? RS - 1 : 1][256][3];
    const int bi = blockIdx.y, kb = blockIdx.x, nkb = gridDim.x;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, rs = threadIdx.x >> 8, tcol = threadIdx.x & 255;
    const int mh = ((m + RS - 1) / RS + GROWS - 1) / GROWS * GROWS;       // rows per slice, whole groups
    const int lbeg = rs * mh, lend = min(m, lbeg + mh);
    const float *__restrict__ P = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + (size_t)bi * m * 3;
    const float *__restrict__ mt = match + (size_t)bi * n * m;
    const int k = kb * 256 + tcol;
    const bool live = k < n;
    const int kc = min(k, n - 1);
    const float px = P[kc * 3 + 0], py = P[kc * 3 + 1], pz = P[kc * 3 + 2];
    const int slot = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);          // value index this lane ends up with
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l0 = lbeg; l0 < lbeg + mh; l0 += GROWS) {   // every slice runs the same number of groups (barriers)
        float v[GROWS];
#pragma unroll
        for (int u = 0; u < GROWS; ++u) {                       // 21 independent coalesced loads in flight
            const int l = min(l0 + u, m - 1);
            v[u] = mt[(size_t)l * n + kc];
        }
        float t[64];
        t[63] = 0.f;
#pragma unroll
        for (int u = 0; u < GROWS; ++u) {
            const int l = min(l0 + u, m - 1);                   // wave-uniform -> scalar loads
            const float dx = px - Q[l * 3 + 0], dy = py - Q[l * 3 + 1], dz = pz - Q[l * 3 + 2];
            const float w = live && l0 + u < lend ? v[u] : 0.f;
            const float c = w * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
            const float ex = dx * c, ey = dy * c, ez = dz * c;
            gx += ex; gy += ey; gz += ez;
            t[u * 3 + 0] = ex; t[u * 3 + 1] = ey; t[u * 3 + 2] = ez;
        }
        red[rs][wave][slot] = lane_sums64(t, lane);
        __syncthreads();
        if (tcol < GROWS * 3) {
            const int u = tcol / 3, c = tcol - u * 3;
            if (l0 + u < lend)
                part2[(((size_t)bi * nkb + kb) * m + l0 + u) * 3 + c] =
                    -((red[rs][0][tcol] + red[rs][1][tcol]) + (red[rs][2][tcol] + red[rs][3][tcol]));
        }
        __syncthreads();
    }
    if (RS > 1) {                                               // grad1: the slices' partial sums, added in slice order
        if (rs > 0) { g1s[rs - 1][tcol][0] = gx; g1s[rs - 1][tcol][1] = gy; g1s[rs - 1][tcol][2] = gz; }
        __syncthreads();
        if (rs > 0) return;
#pragma unroll
        for (int q = 0; q + 1 < RS; ++q) { gx += g1s[q][tcol][0]; gy += g1s[q][tcol][1]; gz += g1s[q][tcol][2]; }
    }
    if (live) {
        float *g = grad1 + ((size_t)bi * n + k) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
}

__global__ __launch_bounds__(256) void emd_grad2_sum_kernel(int m, int nkb, const float *__restrict__ part2, float *__restrict__ grad2) {
    const int bi = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * 3) return;
    float s = 0.f;
    for (int kb = 0; kb < nkb; ++kb) s += part2[((size_t)bi * nkb + kb) * m * 3 + i];
    grad2[(size_t)bi * m * 3 + i] = s;
}

// inner-loop slices per workgroup so that the launch has >= ~2048 waves
// slices for the approxmatch passes (both paths use the same ones, so their sums associate identically): enough
// that the deferred kernels (128 points per wave) put ~4 waves on every SIMD
int pick_match_slices(int b, int npoints, int ninner) {
    const long groups = (long)b * ((npoints + PPW - 1) / PPW);
    int s = 1;
    while (s < MAXS && groups * s < 4096 && ninner / (2 * s) >= 64) s *= 2;
    return s;
}

int pick_slices(int b, int npoints, int ninner) {
    const long groups = (long)b * ((npoints + 63) / 64);
    int s = 1;
    while (s < MAXS && groups * s < 2048 && ninner / (2 * s) >= 64) s *= 2;
    return s;
}

}  // namespace

// bytes of the deferred path's own regions (ratio slots, packed-VALU records, materialisation records, cost partials)
static size_t deferred_bytes(int b, int n, int m) {
    // NLEVEL ratio slots | 16 B alignment slack | (x,y,z,w) records of the passes | 12-float records of the materialisation
    return (size_t)b * NLEVEL * ((size_t)n + m) * sizeof(float) + 16 + (size_t)b * ((size_t)n + 2 * (size_t)m) * sizeof(float4) +
           (size_t)b * m * 12 * sizeof(float) + (size_t)b * ((n + PPW - 1) / PPW) * MAXS * sizeof(float);
}
// ... and of the matrix-core passes behind them (MfmaRegions below lays them out; every region 64-byte aligned)
struct MfmaRegions {
    size_t flag, meta, count, counts, recB1, recA2d, recA2[2], idx2[2], remR[2], ratioR, ratioL, order, recAs, rrs, c2s, ls, end;
};
static MfmaRegions mfma_regions(int b, int n, int m) {
    const size_t NP = (size_t)(n + MPW - 1) / MPW * MPW, MP = (size_t)(m + MPW - 1) / MPW * MPW, B = (size_t)b;
    MfmaRegions r;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t here = at; at += (bytes + 63) & ~(size_t)63; return here; };
    constexpr size_t RB = 16 * RECQ;                                             // bytes of a point record
    r.flag = take(16); r.meta = take(B * METAF * 4); r.count = take(B * 4); r.counts = take(NLEVEL * B * 4);
    r.recB1 = take(B * NP * RB); r.recA2d = take(B * MP * RB);
    for (int u = 0; u < 2; ++u) { r.recA2[u] = take(B * MP * RB); r.idx2[u] = take(B * MP * 4); r.remR[u] = take(B * MP * 4); }
    r.ratioR = take(B * MP * 4); r.ratioL = take(B * NP * 4); r.order = take(B * MP * 4);
    r.recAs = take(B * MP * RB); r.rrs = take(NLEVEL * B * MP * 4); r.c2s = take(3 * B * MP * 4); r.ls = take(B * MP * 4);
    r.end = at;
    return r;
}
static size_t mfma_bytes(int b, int n, int m) { return 64 + mfma_regions(b, n, m).end; }

extern "C" size_t dpf_approxmatch_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return deferred_bytes(b, n, m) + mfma_bytes(b, n, m);
}

// 1 (default): the deferred path's passes run on the matrix cores when the coordinates allow it; 0: always the packed-VALU
// kernels (bit-identical to the read-modify-write path).  Returns the previous setting.  Env DPF_EMD_MATRIX=0 sets the default.
extern "C" int dpf_emd_set_matrix_path(int on) {
    const int prev = g_matrix_path ? 1 : 0;
    g_matrix_path = on != 0;
    return prev;
}

static int approxmatch_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp, float *cost,
                            void *workspace, size_t workspace_bytes, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !match || !temp) return DPF_EINVAL;
    if (b > 65535) return DPF_ENOSUP;
    const bool deferred = workspace != nullptr && workspace_bytes >= dpf_approxmatch_workspace_bytes(b, n, m);
    hipStream_t s = (hipStream_t)stream;
    float multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (float)(n / m); }   // integer division, approxmatch.cu:6-12
    else        { multiL = (float)(m / n); multiR = 1; }
    static const bool env_matrix = [] { const char *e = getenv("DPF_EMD_MATRIX"); return !(e && e[0] == '0'); }();
    const bool matrix = deferred && g_matrix_path && env_matrix;
    // regions of the matrix-core passes (behind the deferred path's own)
    const int NP = (n + MPW - 1) / MPW * MPW, MP = (m + MPW - 1) / MPW * MPW;
    uint8_t *mbase = deferred ? (uint8_t *)(((uintptr_t)workspace + deferred_bytes(b, n, m) + 63) & ~(uintptr_t)63) : nullptr;
    const MfmaRegions mr = mfma_regions(b, n, m);
    unsigned *flag = matrix ? (unsigned *)(mbase + mr.flag) : nullptr;
    float *meta = (float *)(mbase + mr.meta);
    hipLaunchKernelGGL(emd_init_kernel, dim3((n + m + 255) / 256, b), dim3(256), 0, s, n, m, multiL, multiR, temp, flag);
    const int s1 = pick_match_slices(b, n, m), s2 = pick_match_slices(b, m, n);
    const dim3 g1((n + 63) / 64, b), g2((m + 63) / 64, b);
    const dim3 h1((n + PPW - 1) / PPW, b), h2((m + PPW - 1) / PPW, b);               // deferred kernels: 128 points per wave
    const size_t rstride = deferred ? (size_t)(n + m) : (size_t)(n + m) * 2;     // per-cloud stride of a ratio slot
    const size_t lstride = (size_t)b * (n + m);                                   // per-level stride in the workspace
    // packed records of the deferred path, after the NLEVEL ratio slots
    const size_t pstride = (size_t)n + 2 * (size_t)m;
    float4 *pk = deferred ? (float4 *)(((uintptr_t)((float *)workspace + NLEVEL * lstride) + 15) & ~(uintptr_t)15) : nullptr;
    float *rec = deferred ? (float *)(pk + (size_t)b * pstride) : nullptr;
    // (the range check first: every kernel of either family looks at its verdict)
    if (matrix) hipLaunchKernelGGL(emd_mfma_prep_kernel, dim3(b), dim3(1024), 0, s, n, m, xyz1, xyz2, meta, flag);
    // (r06, VERDICT r05 #4: the family that is not wanted was moved to a forked side stream -- built, parity-green, measured at
    // cfg5: 2.7323 ms against 2.7315 ms with everything on the caller's stream.  The 31 gated launches sit at the END of the
    // call's sequence, back to back, and their launch latencies overlap each other; r05's "0.13 ms of empty launches" was the
    // rocprof sum of their durations, not time on the call's critical path.  Not kept.)
    if (deferred)
        hipLaunchKernelGGL(emd_pack_init_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, m, multiR, xyz2, pk + n, pstride,
                           (const unsigned *)flag);
    int mfma_cost_parts = 0;
    if (matrix) {
        // the matrix-core family: gate 0.  Per level: pass 1, pass 2, pass 3 over the live list, then its compaction
        // debug (tests/diag/emd_bisect.py): DPF_EMD_STOP_AFTER=k returns behind the k-th launch of this family
        const char *stop_env = getenv("DPF_EMD_STOP_AFTER");
        int stop_left = stop_env ? atoi(stop_env) : -1;
#define EMD_STOP_CHECK() do { if (stop_left > 0 && --stop_left == 0) return (int)hipGetLastError(); } while (0)
        MfmaState st{};
        st.n = n; st.m = m; st.NP = NP; st.MP = MP; st.nb = b;
        st.recB1 = (const u4 *)(mbase + mr.recB1);
        for (int u = 0; u < 2; ++u) {
            st.recA2[u] = (u4 *)(mbase + mr.recA2[u]); st.idx2[u] = (int *)(mbase + mr.idx2[u]); st.remainR_c[u] = (float *)(mbase + mr.remR[u]);
        }
        st.ratioR_c = (float *)(mbase + mr.ratioR); st.ratioL_p = (float *)(mbase + mr.ratioL);
        st.count = (int *)(mbase + mr.count); st.counts = (int *)(mbase + mr.counts); st.order = (int *)(mbase + mr.order);
        st.temp = temp; st.rstride = rstride; st.gate = flag;
        u4 *recA2_dense = (u4 *)(mbase + mr.recA2d), *recA_s = (u4 *)(mbase + mr.recAs);
        float *rr_s = (float *)(mbase + mr.rrs), *c2soa_s = (float *)(mbase + mr.c2s);
        int *l_s = (int *)(mbase + mr.ls);
        float *ws = (float *)workspace;
        hipLaunchKernelGGL(emd_mfma_pack_kernel, dim3((NP + MP + 255) / 256, b), dim3(256), 0, s, st, multiR, xyz1, xyz2,
                           (const float *)meta, recA2_dense, ws, lstride);
        EMD_STOP_CHECK();
        const int m1 = pick_mfma_slices(b, n, m), m2 = pick_mfma_slices(b, m, n);
        const dim3 q1(NP / MPW, b), q2(MP / MPW, b);
        // level j's 4^(j-7) as fp16 powers of two (level_vectors): F down to 2^-14 (a normal fp16 number), f the rest, h = 2^(j-7)
        auto h16 = [](int e) { return (unsigned)((e + 15) << 10); };            // bits of the fp16 number 2^e, -14 <= e <= 15
        auto fac_of = [&](int j) { const int e = 2 * (j - 7), eF = e < -14 ? -14 : e; return LevelFac{h16(eF), h16(e - eF), h16(j - 7)}; };
        LevelScales ls;
        int cur = 0, lj = 0;
        for (int j = 7; j > -2; --j, ++lj) {
            float *rb = ws + lj * lstride;
            ls.lf[lj] = fac_of(j);
            hipLaunchKernelGGL(emd_mfma_cols_kernel<0>, q1, dim3(64, m1), 0, s, st, cur, fac_of(j), rb);
        EMD_STOP_CHECK();
            if (j == 7) {
                hipLaunchKernelGGL(emd_mfma_rows_kernel<4>, q2, dim3(64, m2), 0, s, st, cur, fac_of(j), rb, 0);
                EMD_STOP_CHECK();
            } else {
                // (r06: both regimes as ONE launch -- every fourth workgroup of the sparse regime's grid taking four tiles in the
                // dense one -- was built and measured: cfg5 2.73 -> 3.53 ms.  The two launches stay; the one whose regime it is
                // not costs 5-7 us)
                hipLaunchKernelGGL(emd_mfma_rows_kernel<4>, q2, dim3(64, m2), 0, s, st, cur, fac_of(j), rb, 1);
                EMD_STOP_CHECK();
                hipLaunchKernelGGL(emd_mfma_rows_kernel<1>, dim3(MP / 32, b), dim3(64, MSL), 0, s, st, cur, fac_of(j), rb, 2);
                EMD_STOP_CHECK();
            }
            hipLaunchKernelGGL(emd_mfma_cols_kernel<3>, q1, dim3(64, m1), 0, s, st, cur, fac_of(j), rb);
        EMD_STOP_CHECK();
            if (j > -1) {
                hipLaunchKernelGGL(emd_mfma_compact_kernel<false>, dim3(b), dim3(1024), 0, s, st, cur, lj);
        EMD_STOP_CHECK();
                cur ^= 1;
            } else {
                hipLaunchKernelGGL(emd_mfma_compact_kernel<true>, dim3(b), dim3(1024), 0, s, st, cur, lj);
        EMD_STOP_CHECK();
            }
        }
        hipLaunchKernelGGL(emd_mfma_gather_kernel, dim3((MP + 255) / 256, b), dim3(256), 0, s, st, (const u4 *)recA2_dense, xyz2,
                           (const float *)ws, lstride, recA_s, rr_s, c2soa_s, l_s);
        EMD_STOP_CHECK();
        const dim3 qm(NP / (32 * MTM), b);
        float *costpart_m = rec + (size_t)b * m * 12;
        if (cost)
            hipLaunchKernelGGL(emd_mfma_materialize_kernel<true>, qm, dim3(64, m1), 0, s, st, ls, xyz1, (const float *)ws, lstride,
                               (const u4 *)recA_s, (const float *)rr_s, (const float *)c2soa_s, (const int *)l_s, match, costpart_m);
        else
            hipLaunchKernelGGL(emd_mfma_materialize_kernel<false>, qm, dim3(64, m1), 0, s, st, ls, xyz1, (const float *)ws, lstride,
                               (const u4 *)recA_s, (const float *)rr_s, (const float *)c2soa_s, (const int *)l_s, match, costpart_m);
        mfma_cost_parts = (int)qm.x * m1;
#undef EMD_STOP_CHECK
    }
    Levels lv;
    int li = 0;
    for (int j = 7; j > -2; --j, ++li) {                    // approxmatch.cu:24 (the j==-2 branch is dead)
        const float level = -powf(4.0f, (float)j);
        const float lvl2 = level * 1.44269504088896340736f;  // exp(x) = exp2(x*log2 e), as __expf does
        lv.lvl2[li] = lvl2;
        float *rb = deferred ? (float *)workspace + li * lstride : temp + (size_t)(n + m);
        if (deferred) {
            // the packed-VALU family: gate 1 (always when the matrix path is off: flag == nullptr)
            hipLaunchKernelGGL(emd_ratio2_kernel<1>, h1, dim3(64, s1), 0, s, n, m, lvl2, xyz1, xyz2, temp, rb, rstride, pk, pstride,
                               (const unsigned *)flag);
            hipLaunchKernelGGL(emd_ratio2_kernel<2>, h2, dim3(64, s2), 0, s, n, m, lvl2, xyz1, xyz2, temp, rb, rstride, pk, pstride,
                               (const unsigned *)flag);
            hipLaunchKernelGGL(emd_match2_kernel, h1, dim3(64, s1), 0, s, n, m, lvl2, xyz1, temp, rb, rstride, (const float4 *)pk, pstride,
                               (const unsigned *)flag);
            continue;
        }
        hipLaunchKernelGGL(emd_ratio_kernel<1>, g1, dim3(64, s1), 0, s, n, m, lvl2, xyz1, xyz2, temp, rb, rstride);
        hipLaunchKernelGGL(emd_ratio_kernel<2>, g2, dim3(64, s2), 0, s, n, m, lvl2, xyz1, xyz2, temp, rb, rstride);
        if (j == 7)
            hipLaunchKernelGGL(emd_match_kernel<0>, g1, dim3(64, s1), 0, s, n, m, lvl2, xyz1, xyz2, match, temp, rb, rstride);
        else
            hipLaunchKernelGGL(emd_match_kernel<1>, g1, dim3(64, s1), 0, s, n, m, lvl2, xyz1, xyz2, match, temp, rb, rstride);
    }
    if (deferred) {
        LevelPairs lp;
        for (int i = 0; i < (NLEVEL + 1) / 2; ++i) {
            uint32_t lo, hi = 0;
            memcpy(&lo, &lv.lvl2[2 * i], 4);
            if (2 * i + 1 < NLEVEL) memcpy(&hi, &lv.lvl2[2 * i + 1], 4);
            lp.p[i] = (u64)lo | ((u64)hi << 32);
        }
        hipLaunchKernelGGL(emd_pack_levels_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, n, m, xyz2,
                           (const float *)workspace, lstride, rstride, rec, (const unsigned *)flag);
        float *costpart = rec + (size_t)b * m * 12;
        if (cost) {
            hipLaunchKernelGGL(emd_materialize2_kernel<true>, h1, dim3(64, s1), 0, s, n, m, lp, xyz1, (const float *)rec, match,
                               (const float *)workspace, lstride, rstride, costpart, (const unsigned *)flag);
            hipLaunchKernelGGL(emd_cost_sum_kernel, dim3(b), dim3(256), 0, s, (int)h1.x * s1, (const float *)costpart, cost,
                               (const unsigned *)flag, 1u);
            if (matrix)
                hipLaunchKernelGGL(emd_cost_sum_kernel, dim3(b), dim3(256), 0, s, mfma_cost_parts, (const float *)costpart, cost,
                                   (const unsigned *)flag, 0u);
        } else {
            hipLaunchKernelGGL(emd_materialize2_kernel<false>, h1, dim3(64, s1), 0, s, n, m, lp, xyz1, (const float *)rec, match,
                               (const float *)workspace, lstride, rstride, costpart, (const unsigned *)flag);
        }
    }
    return (int)hipGetLastError();
}

// Test hook (tests/test_gpu_emd.py::test_matrix_core_exponent_error_bound): out (m, n) = the matrix-core passes' exp2 arguments
// -4^j log2(e) |x2_l - x1_k|^2 of ONE cloud pair at level j (7 .. -1); meta8 (device, 8 floats, may be NULL) receives the
// centroid, the out-of-range flag (then `out` is left untouched), 2^g and T.  Workspace as for dpf_approxmatch_ws(1, n, m).
extern "C" int dpf_debug_emd_exponents(int n, int m, const float *xyz1, const float *xyz2, int level_j, float *out, float *meta8,
                                       void *workspace, size_t workspace_bytes, dpf_stream_t stream) {
    static const unsigned slot_mask = [] { const char *e = getenv("DPF_EMD_DEBUG_SLOTS"); return e ? (unsigned)strtoul(e, nullptr, 0) : ~0u; }();
    if (n <= 0 || m <= 0 || level_j < -1 || level_j > 7 || !xyz1 || !xyz2 || !out) return DPF_EINVAL;
    if (!workspace || workspace_bytes < dpf_approxmatch_workspace_bytes(1, n, m)) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int NP = (n + MPW - 1) / MPW * MPW, MP = (m + MPW - 1) / MPW * MPW;
    uint8_t *mbase = (uint8_t *)(((uintptr_t)workspace + deferred_bytes(1, n, m) + 63) & ~(uintptr_t)63);
    const MfmaRegions mr = mfma_regions(1, n, m);
    unsigned *flag = (unsigned *)(mbase + mr.flag);
    float *meta = (float *)(mbase + mr.meta);
    hipError_t e = dpf_zero_async(flag, 16, s);
    if (e != hipSuccess) return (int)e;
    MfmaState st{};
    st.n = n; st.m = m; st.NP = NP; st.MP = MP; st.nb = 1;
    st.recB1 = (const u4 *)(mbase + mr.recB1);
    for (int u = 0; u < 2; ++u) {
        st.recA2[u] = (u4 *)(mbase + mr.recA2[u]); st.idx2[u] = (int *)(mbase + mr.idx2[u]); st.remainR_c[u] = (float *)(mbase + mr.remR[u]);
    }
    st.ratioR_c = (float *)(mbase + mr.ratioR); st.ratioL_p = (float *)(mbase + mr.ratioL);
    st.count = (int *)(mbase + mr.count); st.counts = (int *)(mbase + mr.counts); st.order = (int *)(mbase + mr.order);
    st.temp = nullptr; st.rstride = (size_t)(n + m); st.gate = flag;
    u4 *recA2_dense = (u4 *)(mbase + mr.recA2d);
    hipLaunchKernelGGL(emd_mfma_prep_kernel, dim3(1), dim3(1024), 0, s, n, m, xyz1, xyz2, meta, flag);
    hipLaunchKernelGGL(emd_mfma_pack_kernel, dim3((NP + MP + 255) / 256, 1), dim3(256), 0, s, st, 1.0f, xyz1, xyz2, (const float *)meta,
                       recA2_dense, (float *)workspace, (size_t)(n + m));
    const int ex = 2 * (level_j - 7), eF = ex < -14 ? -14 : ex;
    const LevelFac lf{(unsigned)((eF + 15) << 10), (unsigned)((ex - eF + 15) << 10), (unsigned)((level_j - 7 + 15) << 10)};
    hipLaunchKernelGGL(emd_mfma_debug_exponents_kernel, dim3(NP / 32, MP / 32), dim3(64), 0, s, st, lf, (const u4 *)recA2_dense, out, slot_mask);
    if (meta8) {
        e = hipMemcpyAsync(meta8, meta, METAF * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipGetLastError();
}

#if EMD_DBG & 8
extern "C" int dpf_debug_emd_mismatches(float *out129) {     // EMD_DBG & 8 builds: [count, 16 x 8 records]; resets the count
    unsigned n = 0;
    float rec[128];
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_emd_dbg_n), 4) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_emd_dbg), sizeof(rec)) != hipSuccess) return -1;
    out129[0] = (float)n;
    memcpy(out129 + 1, rec, sizeof(rec));
    n = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_emd_dbg_n), &n, 4);
    return 0;
}
#endif

extern "C" int dpf_approxmatch_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                                  void *workspace, size_t workspace_bytes, dpf_stream_t stream) {
    return approxmatch_impl(b, n, m, xyz1, xyz2, match, temp, nullptr, workspace, workspace_bytes, stream);
}

// approxmatch + matchcost in one call: the materialisation pass also sums match * distance (fixed-order partial
// sums: deterministic).  The workspace is required (there is no read-modify-write variant of this entry point).
extern "C" int dpf_approxmatch_cost_ws(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                                       float *cost, void *workspace, size_t workspace_bytes, dpf_stream_t stream) {
    if (!cost || !workspace || (b > 0 && n > 0 && m > 0 && workspace_bytes < dpf_approxmatch_workspace_bytes(b, n, m)))
        return DPF_EINVAL;
    return approxmatch_impl(b, n, m, xyz1, xyz2, match, temp, cost, workspace, workspace_bytes, stream);
}

extern "C" int dpf_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                               dpf_stream_t stream) {
    return dpf_approxmatch_ws(b, n, m, xyz1, xyz2, match, temp, nullptr, 0, stream);
}

extern "C" int dpf_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                             dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !match || !out) return DPF_EINVAL;
    if (b > 65535) return DPF_ENOSUP;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = dpf_zero_async(out, sizeof(float) * b, s);
    if (e != hipSuccess) return (int)e;
    int gx = (int)(2048 / b);
    if (gx < 1) gx = 1;
    if (gx > m) gx = m;
    hipLaunchKernelGGL(emd_cost_kernel, dim3(gx, b), dim3(256), 0, s, n, m, xyz1, xyz2, match, out);
    return (int)hipGetLastError();
}

extern "C" int dpf_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                 float *grad1, float *grad2, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !match || !grad1 || !grad2) return DPF_EINVAL;
    if (b > 65535) return DPF_ENOSUP;
    hipStream_t s = (hipStream_t)stream;
    const int s1 = pick_slices(b, n, m);
    hipLaunchKernelGGL(emd_grad1_kernel, dim3((n + 63) / 64, b), dim3(64, s1), 0, s, n, m, xyz1, xyz2, match, grad1);
    hipLaunchKernelGGL(emd_grad2_kernel, dim3((m + 3) / 4, b), dim3(256), 0, s, n, m, xyz1, xyz2, match, grad2);
    return (int)hipGetLastError();
}

// Same gradients with `match` read ONCE (fixed-order sums: deterministic, but not bit-identical to
// dpf_matchcostgrad, whose sums run in a different order).  NULL / short workspace -> dpf_matchcostgrad.
extern "C" size_t dpf_matchcostgrad_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return (size_t)b * ((n + 255) / 256) * m * 3 * sizeof(float);
}

extern "C" int dpf_matchcostgrad_ws(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                    float *grad1, float *grad2, void *workspace, size_t workspace_bytes,
                                    dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !match || !grad1 || !grad2) return DPF_EINVAL;
    if (b > 65535) return DPF_ENOSUP;
    const int nkb = (n + 255) / 256;
    // one workgroup per 256 columns and cloud: with fewer than one per CU the row-parallel two-pass kernels win
    // (measured r01, one pass vs two kernels: B=2, N=8192: 0.58 vs 0.29 ms; B=32, N=2048: 0.37 vs 0.30 ms;
    // B=16, N=8192: 1.52 vs 2.26 ms)
    if (!workspace || workspace_bytes < dpf_matchcostgrad_workspace_bytes(b, n, m) || (long)b * nkb < 512)
        return dpf_matchcostgrad(b, n, m, xyz1, xyz2, match, grad1, grad2, stream);
    hipStream_t s = (hipStream_t)stream;
    if ((long)b * nkb < 1024 && m >= 4 * GROWS)     // fewer than four column-waves per SIMD: two row slices per workgroup
        hipLaunchKernelGGL(emd_grad_fused_kernel<2>, dim3(nkb, b), dim3(512), 0, s, n, m, xyz1, xyz2, match, grad1, (float *)workspace);
    else
        hipLaunchKernelGGL(emd_grad_fused_kernel<1>, dim3(nkb, b), dim3(256), 0, s, n, m, xyz1, xyz2, match, grad1, (float *)workspace);
    hipLaunchKernelGGL(emd_grad2_sum_kernel, dim3((m * 3 + 255) / 256, b), dim3(256), 0, s, m, nkb, (const float *)workspace, grad2);
    return (int)hipGetLastError();
}
