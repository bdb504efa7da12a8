// Exact Chamfer nearest neighbour, matrix-core filtered, for gfx950 (MI355X).
//
// Contract = nn_kernel in chamfer.hip: bit-exact d = (dx*dx + dy*dy) + dz*dz
// (no FMA) and the LOWEST index among exact ties, i.e. the result of the
// reference's strict '<' scan in ascending k (nndistance.cu:16-119).
//
// The brute-force kernel spends ~5.4 VALU instructions per (query, candidate)
// pair and is bound by VALU issue.  Here the matrix cores compute, for every
// pair, the SURROGATE  s(q,c) = |c|^2 - 2 q.c  (= d - |q|^2) in one
// v_mfma_f32_32x32x16_bf16 per 32x32 pairs: the 16 K-slots hold the hi/lo bf16
// split of the candidate (cx,cy,cz) against the hi/lo split of -2q (4 product
// terms per coordinate) and a 3-way split of |c|^2 against 1.  s differs from
// the true distance by at most E = 2^-14 * R2 (R2 = largest squared norm of the
// two clouds; derivation in DESIGN.md 4.3), so the fp32-exact minimiser -- and
// every exact tie -- has s <= s_min + tau with tau = 2^-12 * R2.  ONE sweep over
// the candidate tiles: per tile one MFMA and a v_min3 tree over the accumulator
// fragment (0.5 VALU op per pair); a tile whose minimum is within tau of the
// RUNNING minimum is queued per lane (the running minimum only decreases, so the
// queue is a superset of what the final threshold selects: ~ln(#tiles) record
// lows plus near-ties).  Afterwards the queue is filtered with the final s_min
// and only those few tiles are evaluated with the exact formula and the
// (d, index) lexicographic rule.  A queue overflow (pathological data: thousands
// of near-ties) makes the wave rescan everything exactly, so the result is exact
// for any finite input.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "dpf_hip.h"

#pragma clang fp contract(off)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int QW = 16;           // waves per workgroup, 32 queries each
constexpr int CT = 64;           // candidate tiles resident in LDS at a time (64 KiB of fragments + 32 KiB of points)
constexpr int QCAP = 8;          // queued candidate tiles per lane (compacted when full)

__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t bf16_rne(float x) {
    const uint32_t u = f2u(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split2(float x, uint32_t &hi, uint32_t &lo) {   // x ~ hi + lo, both bf16
    const uint32_t h = f2u(x) & 0xFFFF0000u;
    hi = h >> 16;
    lo = bf16_rne(x - u2f(h));
}

__host__ __device__ inline int tiles_of(int n) { return (n + 31) / 32; }

// ---- prep: per cloud, the MFMA fragments of its points in both roles + the cloud's max |p|^2 ----
// fragment of a 32-point tile: lane (i = point in tile, h) holds K slots 8h..8h+7 (16 bytes)
//   slots: [x:0-3] [y:4-7] [z:8-11] [norm:12-14] [15: 0]
//   role A (candidate): coord -> (ch, ch, cl, cl),   norm -> (wh, wm, wl)      w = (x*x + y*y) + z*z
//   role B (query):     coord -> (qh, ql, qh, ql) of -2q,  norm -> (1, 1, 1)
struct PrepSet {
    const float *xyz;   // (B, n, 3)
    uint4 *fa, *fb;     // (B, ntiles, 64)
    float *tmax;        // (B, ntiles) largest |p|^2 of each tile
    float4 *pts;        // (B, ntiles*32) points as (x, y, z, 0): one 16-byte load per exact evaluation
    int n;
};
struct PrepArgs { PrepSet s[2]; };

__global__ __launch_bounds__(256) void nnm_prep_kernel(PrepArgs args) {
    const PrepSet S = args.s[blockIdx.z];
    const int bi = blockIdx.y, n = S.n;
    const int gid = blockIdx.x * 256 + threadIdx.x;            // one thread per (tile, lane)
    const int tile = gid >> 6, lane = gid & 63, i = lane & 31, h = lane >> 5;
    if (tile >= tiles_of(n)) return;
    const int p = tile * 32 + i;
    float x = 0.f, y = 0.f, z = 0.f;
    const bool live = p < n;
    if (live) {
        const float *src = S.xyz + ((size_t)bi * n + p) * 3;
        x = src[0]; y = src[1]; z = src[2];
    }
    const float w = (x * x + y * y) + z * z;
    uint32_t sa[16], sb[16];
    const float c3[3] = {x, y, z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t ch, cl, qh, ql;
        split2(c3[c], ch, cl);
        split2(-2.0f * c3[c], qh, ql);
        sa[4 * c + 0] = ch; sa[4 * c + 1] = ch; sa[4 * c + 2] = cl; sa[4 * c + 3] = cl;
        sb[4 * c + 0] = qh; sb[4 * c + 1] = ql; sb[4 * c + 2] = qh; sb[4 * c + 3] = ql;
    }
    {
        // padding rows of the last tile: +big surrogate so they never reach a minimum
        const float ww = live ? w : 3.0e38f;
        const uint32_t wh = f2u(ww) & 0xFFFF0000u;
        const float r1 = ww - u2f(wh);
        const uint32_t wm = f2u(r1) & 0xFFFF0000u;
        const float r2 = r1 - u2f(wm);
        sa[12] = wh >> 16; sa[13] = live ? (wm >> 16) : 0u; sa[14] = live ? bf16_rne(r2) : 0u; sa[15] = 0u;
        sb[12] = 0x3F80u; sb[13] = 0x3F80u; sb[14] = 0x3F80u; sb[15] = 0u;
    }
    uint4 oa, ob;
    const int o = 8 * h;
    oa.x = sa[o + 0] | (sa[o + 1] << 16); oa.y = sa[o + 2] | (sa[o + 3] << 16);
    oa.z = sa[o + 4] | (sa[o + 5] << 16); oa.w = sa[o + 6] | (sa[o + 7] << 16);
    ob.x = sb[o + 0] | (sb[o + 1] << 16); ob.y = sb[o + 2] | (sb[o + 3] << 16);
    ob.z = sb[o + 4] | (sb[o + 5] << 16); ob.w = sb[o + 6] | (sb[o + 7] << 16);
    const size_t off = ((size_t)bi * tiles_of(n) + tile) * 64 + lane;
    S.fa[off] = oa;
    S.fb[off] = ob;
    float m = live ? w : 0.f;
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) S.tmax[(size_t)bi * tiles_of(n) + tile] = m;
    if (h == 0) S.pts[((size_t)bi * (tiles_of(n) + 1) + tile) * 32 + i] = make_float4(x, y, z, 0.f);   // +1 tile of slack per cloud
}

struct MDir {
    const float4 *qp, *cp;     // packed points of the query / candidate cloud
    const uint4 *qfb, *cfa;    // query fragments (role B), candidate fragments (role A)
    const float *qtmax, *ctmax;
    float *dist;
    int *idx;
    int nq, nc;
};
struct MArgs { MDir d[2]; int debug; };

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), z, 0, 0, 0);
}
__device__ __forceinline__ float tile_min(const f32x16 &s) {
    float m = fminf(fminf(s[0], s[1]), s[2]);
    m = fminf(fminf(m, s[3]), s[4]);   m = fminf(fminf(m, s[5]), s[6]);
    m = fminf(fminf(m, s[7]), s[8]);   m = fminf(fminf(m, s[9]), s[10]);
    m = fminf(fminf(m, s[11]), s[12]); m = fminf(fminf(m, s[13]), s[14]);
    return fminf(m, s[15]);
}
__device__ __forceinline__ float dist3(float cx, float cy, float cz, float qx, float qy, float qz) {
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}
// exact evaluation of the 16 candidates this lane half sees in candidate tile `t`
// (accumulator rows (r&3) + 8(r>>2) + 4h); ascending k, so ties keep the lowest index
template <class P>
__device__ __forceinline__ void exact_tile(P cp, int nc, int t, int tl, int h, float qx, float qy, float qz,
                                           float &best, int &bidx) {   // tl = tile index inside cp, t = global tile
    float4 v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] = cp[(size_t)tl * 32 + 8 * g + 4 * h + e];    // padded: always in range
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = t * 32 + 8 * g + 4 * h + e;
            const float d = dist3(v[4 * g + e].x, v[4 * g + e].y, v[4 * g + e].z, qx, qy, qz);
            const bool better = k < nc && (d < best || (d == best && k < bidx));
            best = better ? d : best;
            bidx = better ? k : bidx;
        }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// A workgroup = 16 waves x 32 queries of one cloud.  The candidate cloud's MFMA fragments and
// packed points are DMA'd into LDS once (global_load_lds, 64 tiles = 96 KiB per pass) and shared
// by all 16 waves; per tile a wave issues one ds_read_b128, one MFMA and a v_min3 tree.
__global__ __launch_bounds__(QW * 64) void nnm_kernel(MArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint4 *sfrag = (uint4 *)lds;                                  // [CT][64]
    float4 *spts = (float4 *)(lds + CT * 1024);                    // [CT][32]
    unsigned short *qtile = (unsigned short *)(lds + CT * 1536);   // [QW][QCAP][64]
    float *qmin = (float *)(lds + CT * 1536 + QW * QCAP * 64 * 2); // [QW][QCAP][64]
    const MDir A = args.d[blockIdx.z];
    const int bi = blockIdx.y;
    const int nq = A.nq, nc = A.nc;
    if ((int)blockIdx.x * QW * 32 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = blockIdx.x * QW + wave;                                   // query tile of this wave
    const int nqt = tiles_of(nq), nct = tiles_of(nc);
    const bool wave_live = qt < nqt;
    const float4 *__restrict__ cp = A.cp + (size_t)bi * (nct + 1) * 32;
    const uint4 *__restrict__ cfa = A.cfa + (size_t)bi * nct * 64;
    const int j = qt * 32 + (lane & 31);
    const float4 qv = A.qp[(size_t)bi * (nqt + 1) * 32 + min(qt, nqt - 1) * 32 + (lane & 31)];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    const uint4 bq = A.qfb[((size_t)bi * nqt + min(qt, nqt - 1)) * 64 + lane];
    // R2 = largest |p|^2 of the two clouds (per-tile maxima from the prep kernel)
    float r2 = 0.f;
    for (int t = lane; t < nqt; t += 64) r2 = fmaxf(r2, A.qtmax[(size_t)bi * nqt + t]);
    for (int t = lane; t < nct; t += 64) r2 = fmaxf(r2, A.ctmax[(size_t)bi * nct + t]);
    for (int d = 32; d > 0; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
    const float tau = r2 * 2.44140625e-4f;                                    // 2^-12 * R2

    unsigned short *myq = qtile + (size_t)wave * QCAP * 64;
    float *mym = qmin + (size_t)wave * QCAP * 64;
    float smin = __builtin_inff();
    int qcount = 0;
    float best = __builtin_inff();
    int bidx = INT_MAX;
    const int npass = (nct + CT - 1) / CT;
    for (int pass = 0; pass < npass; ++pass) {
        const int t0 = pass * CT, tn = min(CT, nct - t0);
        if (pass > 0) __syncthreads();                                        // everyone is done with the previous pass
        // DMA: fragments (1 KiB per tile) and packed points (512 B per tile, two tiles per wave-instruction)
        for (int t = wave; t < tn; t += QW)
            __builtin_amdgcn_global_load_lds((glb_void *)(cfa + (size_t)(t0 + t) * 64 + lane), (lds_void *)(sfrag + t * 64), 16, 0, 0);
        for (int t = wave * 2; t < tn; t += QW * 2)
            __builtin_amdgcn_global_load_lds((glb_void *)(cp + (size_t)(t0 + t) * 32 + lane), (lds_void *)(spts + t * 32), 16, 0, 0);
        __syncthreads();                                                      // drains the DMA (vmcnt) and publishes it
        if (wave_live) {
            auto visit = [&](int t, float m) {
                if (m <= smin + tau) {                                        // record low or near-tie of the running minimum
                    if (qcount == QCAP) {
                        // full: drop what the CURRENT threshold already excludes (the final one is no larger)
                        const float cur = fminf(smin, m) + tau;
                        int keep = 0;
                        for (int e = 0; e < QCAP; ++e) {
                            const float v = mym[e * 64 + lane];
                            if (v <= cur) { myq[keep * 64 + lane] = myq[e * 64 + lane]; mym[keep * 64 + lane] = v; ++keep; }
                        }
                        qcount = keep;
                    }
                    if (qcount < QCAP) { myq[qcount * 64 + lane] = (unsigned short)t; mym[qcount * 64 + lane] = m; ++qcount; }
                    else qcount = QCAP + 1;                                   // still full of near-ties: exact rescan
                }
                smin = fminf(smin, m);
            };
            int t = 0;
            for (; t + 4 <= tn; t += 4) {                                     // four independent MFMAs in flight
                const f32x16 s0 = mfma(sfrag[(t + 0) * 64 + lane], bq), s1 = mfma(sfrag[(t + 1) * 64 + lane], bq);
                const f32x16 s2 = mfma(sfrag[(t + 2) * 64 + lane], bq), s3 = mfma(sfrag[(t + 3) * 64 + lane], bq);
                const float m0 = tile_min(s0), m1 = tile_min(s1), m2 = tile_min(s2), m3 = tile_min(s3);
                visit(t + 0, m0); visit(t + 1, m1); visit(t + 2, m2); visit(t + 3, m3);
            }
            for (; t < tn; ++t) visit(t, tile_min(mfma(sfrag[t * 64 + lane], bq)));
            // exact evaluation of this pass's surviving tiles against the pass-local threshold (the
            // global minimum can only be lower, so this is a superset; extra exact evaluations are harmless)
            const float thr = fminf(smin, __shfl_xor(smin, 32)) + tau;
            if (__builtin_amdgcn_ballot_w64(qcount > QCAP) != 0) {
                for (int tt = 0; tt < tn; ++tt) exact_tile(spts, nc, t0 + tt, tt, h, qx, qy, qz, best, bidx);
            } else {
                int nsurv = 0;
                for (int e = 0; e < QCAP; ++e)
                    if (e < qcount && mym[e * 64 + lane] <= thr) { myq[nsurv * 64 + lane] = myq[e * 64 + lane]; ++nsurv; }
                int smax = nsurv;
                for (int d = 32; d > 0; d >>= 1) smax = max(smax, __shfl_xor(smax, d));
                smax = __builtin_amdgcn_readfirstlane(smax);
                for (int e = 0; e < smax; ++e) {
                    const bool take = e < nsurv;
                    const int tl = take ? myq[e * 64 + lane] : 0;
                    float b2 = best; int i2 = bidx;
                    exact_tile(spts, nc, t0 + tl, tl, h, qx, qy, qz, b2, i2);
                    if (take) { best = b2; bidx = i2; }
                }
            }
            qcount = 0;                                                       // the queue is per pass; smin carries over
        }
    }
    if (!wave_live) return;
    // merge the two lane halves of each query
    const float od = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bidx, 32);
    if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
    if (h == 0 && j < nq) {
        A.dist[(size_t)bi * nq + j] = best;
        A.idx[(size_t)bi * nq + j] = bidx;
    }
}

}  // namespace

extern "C" size_t dpf_nndistance_mfma_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    const size_t t = (size_t)tiles_of(n) + tiles_of(m);
    return (size_t)b * t * (64 * 16 * 2 + 32 * 16 + 4) + (size_t)b * 2 * 32 * 16 + 256;
}

extern "C" int dpf_nndistance_mfma(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                   int *result_i, float *result2, int *result2_i, void *workspace,
                                   size_t workspace_bytes, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    if (b > 65535 || n > 65535 * 32 || m > 65535 * 32 || !workspace ||
        workspace_bytes < dpf_nndistance_mfma_workspace_bytes(b, n, m) || (n < 32 && m < 32))
        return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    hipStream_t s = (hipStream_t)stream;
    const size_t nt = tiles_of(n), mt = tiles_of(m);
    uint4 *fa1 = (uint4 *)workspace;
    uint4 *fb1 = fa1 + (size_t)b * nt * 64;
    uint4 *fa2 = fb1 + (size_t)b * nt * 64;
    uint4 *fb2 = fa2 + (size_t)b * mt * 64;
    float4 *p1 = (float4 *)(fb2 + (size_t)b * mt * 64);
    float4 *p2 = p1 + (size_t)b * (nt + 1) * 32;
    float *tm1 = (float *)(p2 + (size_t)b * (mt + 1) * 32);
    float *tm2 = tm1 + (size_t)b * nt;
    PrepArgs pa;
    pa.s[0] = PrepSet{xyz, fa1, fb1, tm1, p1, n};
    pa.s[1] = PrepSet{xyz2, fa2, fb2, tm2, p2, m};
    const int tmax = (int)(nt > mt ? nt : mt);
    hipLaunchKernelGGL(nnm_prep_kernel, dim3((tmax * 64 + 255) / 256, b, 2), dim3(256), 0, s, pa);
    MArgs ma;
    ma.d[0] = MDir{p1, p2, fb1, fa2, tm1, tm2, result, result_i, n, m};      // nndistance.cu:126
    ma.d[1] = MDir{p2, p1, fb2, fa1, tm2, tm1, result2, result2_i, m, n};    // nndistance.cu:127
    ma.debug = 0;
    const int nmax = n > m ? n : m;
    const int lds = CT * 1536 + QW * QCAP * 64 * 6;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)nnm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(nnm_kernel, dim3((nmax + QW * 32 - 1) / (QW * 32), b, 2), dim3(QW * 64), lds, s, ma);
    return (int)hipGetLastError();
}
