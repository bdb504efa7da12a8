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
// every exact tie -- has s <= s_min + tau with tau = 2^-12 * R2.  Two sweeps:
//   1. s_min per query (v_min3 over the accumulator fragment: 0.5 VALU op/pair);
//   2. candidates tiles whose tile minimum is <= s_min + tau are queued per lane
//      (a handful per query) and only THOSE are evaluated with the exact formula
//      and the (d, index) lexicographic rule.
// A queue overflow (pathological data: thousands of near-ties) makes the wave
// rescan everything exactly, so the result is exact for any finite input.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "dpf_hip.h"

#pragma clang fp contract(off)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int QW = 8;            // waves per workgroup, 32 queries each
constexpr int CHUNK = 16;        // candidate tiles (of 32) staged in LDS per step
constexpr int QCAP = 12;         // queued candidate tiles per lane before the exact-rescan fallback

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
    int n;
};
struct PrepArgs { PrepSet s[2]; unsigned *r2bits; };   // r2bits: (B,) max |p|^2 over both clouds, float bits

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
    // cloud-pair bound R2 (non-negative floats order like their bit patterns)
    float m = live ? w : 0.f;
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) atomicMax(args.r2bits + bi, f2u(m));
}

struct MDir {
    const float *q, *c;        // original clouds (B, nq, 3), (B, nc, 3)
    const uint4 *qfb, *cfa;    // query fragments (role B), candidate fragments (role A)
    float *dist;
    int *idx;
    int nq, nc;
};
struct MArgs { MDir d[2]; const unsigned *r2bits; };

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
__device__ __forceinline__ void exact_tile(const float *__restrict__ c, int nc, int t, int h, float qx, float qy, float qz,
                                           float &best, int &bidx) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = t * 32 + 8 * g + 4 * h + e;          // accumulator row (r&3) + 8(r>>2) + 4h
            if (k < nc) {
                const float d = dist3(c[(size_t)k * 3 + 0], c[(size_t)k * 3 + 1], c[(size_t)k * 3 + 2], qx, qy, qz);
                const bool better = d < best || (d == best && k < bidx);
                best = better ? d : best;
                bidx = better ? k : bidx;
            }
        }
}

__global__ __launch_bounds__(QW * 64) void nnm_kernel(MArgs args) {
    __shared__ __attribute__((aligned(16))) uint4 stage[2][CHUNK * 64];      // 2 x 16 KB of candidate fragments
    __shared__ int queue[QW][QCAP][64];
    const MDir A = args.d[blockIdx.z];
    const int bi = blockIdx.y;
    const int nq = A.nq, nc = A.nc;
    if ((int)blockIdx.x * QW * 32 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = blockIdx.x * QW + wave;                                   // query tile of this wave
    const int nqt = tiles_of(nq), nct = tiles_of(nc);
    const bool wave_live = qt < nqt;
    const float *__restrict__ q = A.q + (size_t)bi * nq * 3;
    const float *__restrict__ c = A.c + (size_t)bi * nc * 3;
    const uint4 *__restrict__ cfa = A.cfa + (size_t)bi * nct * 64;
    const int j = qt * 32 + (lane & 31);
    const int jc = min(j, nq - 1);
    const float qx = q[jc * 3 + 0], qy = q[jc * 3 + 1], qz = q[jc * 3 + 2];
    const uint4 bq = A.qfb[((size_t)bi * nqt + min(qt, nqt - 1)) * 64 + lane];
    const float tau = u2f(args.r2bits[bi]) * 2.44140625e-4f;                  // 2^-12 * R2

    const int nchunk = (nct + CHUNK - 1) / CHUNK;
    auto stage_chunk = [&](int ch, int buf) {                                // all 512 threads copy 16 KB
        for (int e = tid; e < CHUNK * 64; e += QW * 64) {
            const int t = ch * CHUNK + (e >> 6);
            uint4 v = {0u, 0u, 0u, 0u};
            if (t < nct) v = cfa[(size_t)t * 64 + (e & 63)];
            stage[buf][e] = v;
        }
    };
    float smin = __builtin_inff();
    int qcount = 0;
    bool overflow = false;
    // sweep 0: minimum of the surrogate; sweep 1: queue the tiles that can hold the exact minimiser
    for (int sweep = 0; sweep < 2; ++sweep) {
        float thr = 0.f;
        if (sweep == 1) {
            const auto sw = __builtin_amdgcn_permlane32_swap(f2u(smin), f2u(smin), false, false);
            smin = fminf(u2f(sw[0]), u2f(sw[1]));                             // both halves of a query
            thr = smin + tau;
        }
        stage_chunk(0, 0);
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            if (ch + 1 < nchunk) stage_chunk(ch + 1, (ch + 1) & 1);
            const uint4 *sb = stage[ch & 1];
            const int tcount = min(CHUNK, nct - ch * CHUNK);
            if (wave_live) {
#pragma unroll 4
                for (int t = 0; t < tcount; ++t) {
                    const f32x16 s = mfma(sb[t * 64 + lane], bq);
                    const float m = tile_min(s);
                    if (sweep == 0) {
                        smin = fminf(smin, m);
                    } else if (m <= thr) {
                        if (qcount < QCAP) queue[wave][qcount][lane] = ch * CHUNK + t;
                        else overflow = true;
                        ++qcount;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!wave_live) return;
    // exact evaluation of the queued tiles (d, index) lexicographic; both halves then merge
    float best = __builtin_inff();
    int bidx = INT_MAX;
    if (__builtin_amdgcn_ballot_w64(overflow) != 0) {
        for (int t = 0; t < nct; ++t) exact_tile(c, nc, t, h, qx, qy, qz, best, bidx);
    } else {
        int qmax = qcount;
        for (int d = 32; d > 0; d >>= 1) qmax = max(qmax, __shfl_xor(qmax, d));
        qmax = __builtin_amdgcn_readfirstlane(qmax);
        for (int e = 0; e < qmax; ++e) {
            if (e < qcount) exact_tile(c, nc, queue[wave][e][lane], h, qx, qy, qz, best, bidx);
        }
    }
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
    return (size_t)b * ((size_t)tiles_of(n) + tiles_of(m)) * 64 * 16 * 2 + (size_t)b * 4 + 256;
}

extern "C" int dpf_nndistance_mfma(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                   int *result_i, float *result2, int *result2_i, void *workspace,
                                   size_t workspace_bytes, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    if (b > 65535 || !workspace || workspace_bytes < dpf_nndistance_mfma_workspace_bytes(b, n, m) || (n < 32 && m < 32))
        return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    hipStream_t s = (hipStream_t)stream;
    const int nt = tiles_of(n), mt = tiles_of(m);
    uint4 *fa1 = (uint4 *)workspace;
    uint4 *fb1 = fa1 + (size_t)b * nt * 64;
    uint4 *fa2 = fb1 + (size_t)b * nt * 64;
    uint4 *fb2 = fa2 + (size_t)b * mt * 64;
    unsigned *r2 = (unsigned *)(fb2 + (size_t)b * mt * 64);
    hipError_t e = hipMemsetAsync(r2, 0, sizeof(unsigned) * b, s);
    if (e != hipSuccess) return (int)e;
    PrepArgs pa;
    pa.s[0] = PrepSet{xyz, fa1, fb1, n};
    pa.s[1] = PrepSet{xyz2, fa2, fb2, m};
    pa.r2bits = r2;
    const int tmax = nt > mt ? nt : mt;
    hipLaunchKernelGGL(nnm_prep_kernel, dim3((tmax * 64 + 255) / 256, b, 2), dim3(256), 0, s, pa);
    MArgs ma;
    ma.d[0] = MDir{xyz, xyz2, fb1, fa2, result, result_i, n, m};      // nndistance.cu:126
    ma.d[1] = MDir{xyz2, xyz, fb2, fa1, result2, result2_i, m, n};    // nndistance.cu:127
    ma.r2bits = r2;
    const int nmax = n > m ? n : m;
    hipLaunchKernelGGL(nnm_kernel, dim3((nmax + QW * 32 - 1) / (QW * 32), b, 2), dim3(QW * 64), 0, s, ma);
    return (int)hipGetLastError();
}
