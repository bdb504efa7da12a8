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
// two clouds, after centring), so the fp32-exact minimiser -- and every exact tie --
// has s <= s_min + tau with tau = 2^-12 * R2 = 4 E.  Where E comes from: a coordinate
// is hi + lo with hi its top 8 bits (truncated: |x - hi| < 2^-7 |x|) and lo the
// bf16 nearest to the remainder (|x - hi - lo| <= 2^-9 * 2^-7 |x| = 2^-16 |x|), on
// both sides; of the four products of (qh + ql)(ch + cl) all four are in the K slots,
// so what is lost per coordinate is the split's own remainder, <= 2 * 2^-16 |q_k||c_k|
// up to second order, i.e. <= 2^-14 |q.c|-bound per pair on the -2 q.c term:
// 2 * sum_k 2^-15 |q_k||c_k| <= 2^-14 |q||c| <= 2^-14 R2.  |c|^2 is a 3-way split
// (24 bits: exact) and the fp32 accumulation of the 15 products adds a few 2^-24 R2.
// tests/test_gpu_chamfer.py::test_filter_surrogate_error_bound MEASURES |s - (d - |q|^2)|
// against E on the adversarial distributions through dpf_debug_nn_surrogate (below).
// The centring rounds c - mu to 2^-24 of itself: < 2^-21 R2 more, inside tau's slack.  ONE sweep over
// the candidate tiles: per tile one MFMA and a v_min3 tree over the accumulator
// fragment (0.5 VALU op per pair).  Since r04 the sweep keeps no queue: a chunk of
// 32 tiles leaves its surrogate minima in registers, then every tile within tau of
// the minimum SO FAR is marked in a per-lane bit mask -- the final minimum can only
// be lower, so the marked tiles are a superset of what the final threshold selects
// -- and only those few tiles are evaluated with the exact formula and the
// (d, index) lexicographic rule, 64 at a time from one wave-wide work list (a lane
// has 1-2 of them, the wave's maximum is ~4: the per-lane loop of r01-r03 paid a
// whole evaluation per lane and round).  A chunk that is full of genuine near-ties
// (duplicate points, lattice data: more than 256 items per wave) keeps the per-lane
// loop, so the result is exact for any finite input and the cost degrades gracefully.
// The surrogate works on coordinates translated to the mean of the first 64
// candidates: tau then scales with the extent of the data, not with its offset.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "dpf_hip.h"
#include "lds_attr.h"
#include "zero_fill.h"
#include "nn_refscan.h"

#pragma clang fp contract(off)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// waves per workgroup (32 queries each): template parameter QW = 16, or 8 when that is what fills the chip
constexpr int CT = 64;           // candidate tiles resident in LDS at a time (64 KiB of fragments + 25 KiB of points)
// points of a tile in LDS: x[32] | y[32] | z[32] | 4 floats of padding.  The exact evaluations read the tiles their lanes'
// survivors name -- any tiles -- at the same offset inside the tile: r04's rows of 32 float4 (512 B apart) put every one of
// those reads on the same four banks (counters: SQ_LDS_BANK_CONFLICT = a third of SQ_LDS_IDX_ACTIVE).  r05: rows of 100
// floats spread consecutive tiles over all banks (18.5 -> 17.7 us at cfg-2 with padded float4 rows; same box), and the
// structure-of-arrays form makes a lane's 16 candidates 12 quad reads instead of 16.
constexpr int PROW = 100;        // floats
constexpr int SCH = 32;          // tiles per sweep chunk: their surrogate minima stay in registers until the chunk's threshold is known
constexpr int WCAP = SCH * 64 / 8;   // work items of a wave per chunk: their 8-byte results reuse the survivor lists' 2 KiB
constexpr int NNM_PTS = CT * 1024, NNM_LISTS = NNM_PTS + CT * PROW * 4;       // byte offsets: fragments | points | survivor lists / results | work lists
__host__ __device__ constexpr int nnm_lds_bytes(int qw) { return NNM_LISTS + qw * SCH * 64 + qw * WCAP * 2; }

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

// ---- MFMA fragments of a point, built in the kernel itself (a separate pre-pass kernel cost 10 us at cfg-2) ----
// fragment of a 32-point tile: lane (i = point in tile, h) holds K slots 8h..8h+7 (16 bytes)
//   slots: [x:0-3] [y:4-7] [z:8-11] [norm:12-14] [15: 0]
//   role A (candidate): coord -> (ch, ch, cl, cl),   norm -> (wh, wm, wl)      w = (x*x + y*y) + z*z
//   role B (query):     coord -> (qh, ql, qh, ql) of -2q,  norm -> (1, 1, 1)
__device__ __forceinline__ void cand_coord(float c, uint32_t &w0, uint32_t &w1) {      // (ch, ch), (cl, cl)
    uint32_t hi, lo;
    split2(c, hi, lo);
    w0 = hi | (hi << 16);
    w1 = lo | (lo << 16);
}
// both halves of a candidate's fragment row; live = false: padding row with a huge surrogate (never a minimum)
__device__ __forceinline__ void cand_fragment(float x, float y, float z, bool live, uint4 &h0, uint4 &h1) {
    cand_coord(x, h0.x, h0.y);
    cand_coord(y, h0.z, h0.w);
    cand_coord(z, h1.x, h1.y);
    const float w = (x * x + y * y) + z * z;
    const float ww = live ? w : 3.0e38f;
    const uint32_t wh = f2u(ww) & 0xFFFF0000u;
    const float r1 = ww - u2f(wh);
    const uint32_t wm = f2u(r1) & 0xFFFF0000u;
    const float r2 = r1 - u2f(wm);
    h1.z = (wh >> 16) | (live ? wm : 0u);                    // slots 12 (wh), 13 (wm)
    h1.w = live ? bf16_rne(r2) : 0u;                          // slot 14 (wl), slot 15 = 0
}
// the half h of a query's fragment column
__device__ __forceinline__ uint4 query_fragment(float x, float y, float z, int h) {
    uint32_t ah, al, bh, bl;
    split2(-2.0f * (h ? z : x), ah, al);
    split2(-2.0f * y, bh, bl);
    uint4 o;
    o.x = ah | (al << 16); o.y = o.x;                         // (qh, ql, qh, ql)
    o.z = h ? 0x3F803F80u : (bh | (bl << 16));                // h = 1: (1, 1 | 1, 0)
    o.w = h ? 0x00003F80u : o.z;
    return o;
}

struct MDir {
    const float *q, *c;        // (B, nq, 3) queries, (B, nc, 3) candidates
    float *dist;
    int *idx;
    int nq, nc;
    long qstride, cstride;     // floats between consecutive clouds (0 = one cloud broadcast over the batch)
};
// pairwise mode (pn2 > 0): blockIdx.y = j, blockIdx.z = 2 i + direction select the pair (clouds1[i], clouds2[j]); instead of
// per-point distances and indices a workgroup writes the SUM of its queries' distances (fixed order) to
// part[((i * pn2 + j) * 2 + direction) * gridDim.x + blockIdx.x] -- the (N1, N2, n) intermediates never exist
// batch mode with part != nullptr (dpf_nndistance_cd): distances and indices are written AND the workgroup's sum goes to
// part[(bi * 2 + direction) * gridDim.x + blockIdx.x] -- the per-cloud CD reduction then reads a handful of partial sums
// instead of the (B, n) distances
// ticket != nullptr: no finish launch -- every workgroup publishes its sum with an agent-scope store, waits for it, and
// takes a ticket of its pair / cloud (agent-scope atomic add); the LAST arriver reads the 2 * gridDim.x sums back with
// agent-scope loads, adds them in tile order, writes cd and resets the ticket (tickets are zero on entry and on exit).
// Payload and ticket go through device-coherent accesses on both sides (MI355X_MICROARCH.md, "valid forms"): per-XCD L2s
// are not coherent with each other and a plain load could be served a stale line of the previous launch's sums.
struct MArgs { MDir d[2]; int pn2; float *part; unsigned *ticket; float *cd; };

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), z, 0, 0, 0);
}
__device__ __forceinline__ float tile_min(const f32x16 &s) {
    float m = fminf(fminf(s[0], s[1]), s[2]);
    m = fminf(fminf(m, s[3]), s[4]);   m = fminf(fminf(m, s[5]), s[6]);
    m = fminf(fminf(m, s[7]), s[8]);   m = fminf(fminf(m, s[9]), s[10]);
    m = fminf(fminf(m, s[11]), s[12]); m = fminf(fminf(m, s[13]), s[14]);
    // eight v_min3: the last one takes s[0] again rather than being a two-input minimum, whose operands the compiler
    // canonicalises with two extra instructions (it cannot know that an MFMA result is not a signalling NaN)
    return fminf(fminf(m, s[15]), s[0]);
}
__device__ __forceinline__ float dist3(float cx, float cy, float cz, float qx, float qy, float qz) {
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}
// exact evaluation of the 16 candidates this lane half sees in candidate tile `t` (accumulator rows
// (r&3) + 8(r>>2) + 4h).  Minimum first (16 distances + a v_min3 tree); only a tile whose minimum reaches the
// current best pays for the index: the LOWEST k attaining the minimum (descending scan, last hit wins), then the
// (d, index) lexicographic rule.  Padding points sit at 3e38, so their distance is +inf.
template <class P>
__device__ __forceinline__ void exact_tile(P cp, int nc, int t, int tl, int h, float qx, float qy, float qz,
                                           float &best, int &bidx) {   // tl = tile index inside cp, t = global tile
    float d[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {                       // four points at a time: few live registers
        const float *row = cp + (size_t)tl * PROW + 8 * g + 4 * h;               // padded: always in range
        const float4 vx = *(const float4 *)row, vy = *(const float4 *)(row + 32), vz = *(const float4 *)(row + 64);
        d[4 * g + 0] = dist3(vx.x, vy.x, vz.x, qx, qy, qz); d[4 * g + 1] = dist3(vx.y, vy.y, vz.y, qx, qy, qz);
        d[4 * g + 2] = dist3(vx.z, vy.z, vz.z, qx, qy, qz); d[4 * g + 3] = dist3(vx.w, vy.w, vz.w, qx, qy, qz);
    }
    float m = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) m = fminf(fminf(m, d[r]), d[r + 1]);
    m = fminf(m, d[15]);
    if (m <= best) {
        int kmin = INT_MAX;
#pragma unroll
        for (int r = 15; r >= 0; --r) {
            const int k = t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            kmin = (d[r] == m && k < nc) ? k : kmin;
        }
        const bool better = kmin != INT_MAX && (m < best || kmin < bidx);
        best = better ? m : best;
        bidx = better ? kmin : bidx;
    }
}

// the same evaluation for ANOTHER lane's query (the wave-wide work list of nnm_kernel): minimum and lowest index of the 16
// candidates lane half `h` sees in tile `t`, for the owner to merge under the (d, index) rule (INT_MAX: no candidate)
template <class P>
__device__ __forceinline__ void exact_tile_mk(P cp, int nc, int t, int tl, int h, float qx, float qy, float qz, float &m, int &kmin) {
    // r06: one candidate per instruction.  r04-r05 evaluated two per PACKED fp32 instruction here.  On gfx950 a packed fp32 VALU
    // instruction whose low half reads the high word of a VGPR pair loses that half in lanes 48-63 while another wave of the SIMD
    // issues MFMAs at certain distances (tools/ubench/pk_vs_mfma_forms.hip, profiles/r06_packed_f32_vs_mfma.txt; found behind
    // approx-EMD's run-to-run differing bits) -- and the other fifteen waves of this workgroup sweep with MFMAs while
    // this one evaluates.  The forms that stood here were the plain ones, which the micro-test finds immune (and 2 000 repeats +
    // 1.3e5 fuzz cases never differed); the rule of csrc/Makefile is nevertheless the simple one -- a kernel that issues MFMAs keeps
    // no packed fp32 at all (tools/mfma_overlap_check.py --no-packed-with-mfma) -- and it costs nothing here (16.7-17.0 us either way)
    float d[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float *row = cp + (size_t)tl * PROW + 8 * g + 4 * h;
        const float4 vx = *(const float4 *)row, vy = *(const float4 *)(row + 32), vz = *(const float4 *)(row + 64);
        d[4 * g + 0] = dist3(vx.x, vy.x, vz.x, qx, qy, qz); d[4 * g + 1] = dist3(vx.y, vy.y, vz.y, qx, qy, qz);
        d[4 * g + 2] = dist3(vx.z, vy.z, vz.z, qx, qy, qz); d[4 * g + 3] = dist3(vx.w, vy.w, vz.w, qx, qy, qz);
    }
    m = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) m = fminf(fminf(m, d[r]), d[r + 1]);
    m = fminf(m, d[15]);
    kmin = INT_MAX;
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        const int k = t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        kmin = (d[r] == m && k < nc) ? k : kmin;
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// A workgroup = 16 waves x 32 queries of one cloud.  Per pass of 64 candidate tiles the workgroup builds the
// candidates' MFMA fragments and packed points in LDS (96 KiB; two points per thread) and all 16 waves share them;
// per tile a wave issues one ds_read_b128, one MFMA and a v_min3 tree.  R2 is taken over ALL candidates and over
// the queries of THIS workgroup: the surrogate's error bound is per pair.
template <int QW>
__global__ __launch_bounds__(QW * 64) void nnm_kernel(MArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint4 *sfrag = (uint4 *)lds;                                  // [CT][64]
    float *spts = (float *)(lds + NNM_PTS);                        // [CT][PROW]
    unsigned char *qtile = (unsigned char *)(lds + NNM_LISTS);     // [QW][WCAP] 8-byte results of the wave-wide work list
    __shared__ float s_r2[QW];
    const bool pairwise = args.pn2 > 0;
    const int dir = pairwise ? (int)(blockIdx.z & 1) : (int)blockIdx.z;
    const MDir A = args.d[dir];
    // pairwise: direction 0 takes its queries from cloud i of the first set and its candidates from cloud j of the second
    // (strides of the "other" index are 0 in MDir), direction 1 the other way round
    const int pi = pairwise ? (int)(blockIdx.z >> 1) : 0, pj = (int)blockIdx.y;
    const int bi = pairwise ? 0 : (int)blockIdx.y;
    const int nq = A.nq, nc = A.nc;
    const bool sums = args.part != nullptr;                   // pairwise mode always; batch mode on request
    const size_t part_at = pairwise ? (((size_t)pi * args.pn2 + pj) * 2 + dir) * gridDim.x + blockIdx.x
                                    : ((size_t)bi * 2 + dir) * gridDim.x + blockIdx.x;
    // cloud / pair index of the partial sums, and their publication + finish by the last arriver
    const size_t unit = pairwise ? (size_t)pi * args.pn2 + pj : (size_t)bi;
    auto publish = [&](float t) {                                 // called by thread 0 of every workgroup of the launch
        if (args.ticket == nullptr) { args.part[part_at] = t; return; }
        __hip_atomic_store(&args.part[part_at], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the sum has left this CU before the ticket is taken
        const unsigned nwg2 = 2u * gridDim.x;
        const unsigned old = __hip_atomic_fetch_add(&args.ticket[unit], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg2 - 1) {
            const float *p = args.part + unit * nwg2;
            float s1 = 0.f, s2 = 0.f;
            for (unsigned x = 0; x < gridDim.x; ++x) {
                s1 += __hip_atomic_load(&p[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(&p[gridDim.x + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            args.cd[unit] = s1 / (float)args.d[0].nq + s2 / (float)args.d[1].nq;
            __hip_atomic_store(&args.ticket[unit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if ((int)blockIdx.x * QW * 32 >= nq) {
        if (sums && threadIdx.x == 0) publish(0.f);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = blockIdx.x * QW + wave;                                   // query tile of this wave
    const int nqt = tiles_of(nq), nct = tiles_of(nc);
    const bool wave_live = qt < nqt;
    const size_t qcloud = pairwise ? (size_t)(dir == 0 ? pi : pj) * A.qstride : (size_t)bi * A.qstride;
    const size_t ccloud = pairwise ? (size_t)(dir == 0 ? pj : pi) * A.cstride : (size_t)bi * A.cstride;
    const float *__restrict__ cpts = A.c + ccloud;
    const int j = qt * 32 + (lane & 31);
    const float *qsrc = A.q + qcloud + (size_t)min(j, nq - 1) * 3;
    const float qx = qsrc[0], qy = qsrc[1], qz = qsrc[2];
    // The surrogate is evaluated on coordinates translated by mu = the mean of the first 64 candidates (every wave
    // computes the same value: no barrier), so that R2 -- and with it the filter's tolerance tau -- scales with the
    // extent of the data, not with its distance from the origin (an uncentred cloud made every pair a "near tie").
    // Distances are translation invariant and fl(c - mu) is accurate to 2^-24 of ITSELF, which adds < 2^-21 * R2 to
    // the surrogate's error; the exact evaluation keeps using the original coordinates.
    float mux, muy, muz;
    {
        const float *src = cpts + (size_t)min(lane, nc - 1) * 3;
        mux = src[0]; muy = src[1]; muz = src[2];
        for (int d = 32; d > 0; d >>= 1) { mux += __shfl_xor(mux, d); muy += __shfl_xor(muy, d); muz += __shfl_xor(muz, d); }
        mux *= 0.015625f; muy *= 0.015625f; muz *= 0.015625f;
    }
    const float qcx = qx - mux, qcy = qy - muy, qcz = qz - muz;
    const uint4 bq = query_fragment(qcx, qcy, qcz, h);
    // R2 >= |c - mu|^2 of every candidate and |q - mu|^2 of this workgroup's queries (the surrogate's error bound is
    // per pair): the queries now, the candidates while their fragments are built (the first pass covers them all
    // when nc <= 2048; otherwise a pre-scan)
    // Non-finite input (NaN / Inf coordinates, or norms that overflow) poisons R2 with +inf -- fmaxf alone would drop a NaN --
    // and the whole workgroup then takes the reference's own scan (nn_refscan.h) instead of the filter, whose bounds mean
    // nothing there: the result is the reference's for ANY input.
    const float INF_ = __builtin_inff();
    float r2 = j < nq ? (qcx * qcx + qcy * qcy) + qcz * qcz : 0.f;
    r2 = nn_not_finite(r2) ? INF_ : r2;
    if (nct > CT)
        for (int p = tid; p < nc; p += QW * 64) {
            const float *src = cpts + (size_t)p * 3;
            const float x = src[0] - mux, y = src[1] - muy, z = src[2] - muz;
            const float cn = (x * x + y * y) + z * z;
            r2 = nn_not_finite(cn) ? INF_ : fmaxf(r2, cn);
        }
    float tau = 0.f;
    bool slow = false;

    unsigned short *wl = (unsigned short *)(lds + NNM_LISTS + QW * SCH * 64) + (size_t)wave * WCAP;      // (source lane << 8) | tile
    uint2 *res = (uint2 *)(qtile + (size_t)wave * SCH * 64);       // (distance bits, index) per work item of the wave
    float smin = __builtin_inff();
    float best = __builtin_inff();
    int bidx = INT_MAX;
    const int npass = (nct + CT - 1) / CT;
    for (int pass = 0; pass < npass; ++pass) {
        const int t0 = pass * CT, tn = min(CT, nct - t0);
        if (pass > 0) __syncthreads();                                        // everyone is done with the previous pass
        int tidp = tid;                    // opaque: the build's LDS addresses are made here, per pass, not carried through the sweep
        asm volatile("" : "+v"(tidp));
#pragma unroll
        for (int k = 0; k < CT * 32 / (QW * 64); ++k) {                       // two candidates per thread
            const int pl = tidp + k * QW * 64, t = pl >> 5, i = pl & 31, p = t0 * 32 + pl;
            if (t < tn) {
                const bool live = p < nc;
                float x = 0.f, y = 0.f, z = 0.f;
                if (live) { const float *src = cpts + (size_t)p * 3; x = src[0]; y = src[1]; z = src[2]; }
                uint4 f0, f1;
                const float cx = live ? x - mux : 0.f, cy = live ? y - muy : 0.f, cz = live ? z - muz : 0.f;
                cand_fragment(cx, cy, cz, live, f0, f1);
                sfrag[t * 64 + i] = f0;
                sfrag[t * 64 + 32 + i] = f1;
                // original coordinates for the exact evaluation; padding far away (its distance is +inf)
                spts[t * PROW + i] = live ? x : 3.0e38f; spts[t * PROW + 32 + i] = live ? y : 3.0e38f; spts[t * PROW + 64 + i] = live ? z : 3.0e38f;
                if (pass == 0) {
                    const float cn = (cx * cx + cy * cy) + cz * cz;
                    r2 = nn_not_finite(cn) ? INF_ : fmaxf(r2, cn);
                }
            }
        }
        if (pass == 0) {
            for (int d = 32; d > 0; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
            if (lane == 0) s_r2[wave] = r2;
        }
        __syncthreads();                                                      // fragments and points are published
        if (pass == 0) {
            float m = s_r2[0];
#pragma unroll
            for (int w = 1; w < QW; ++w) m = fmaxf(m, s_r2[w]);
            tau = m * 2.44140625e-4f;                                         // 2^-12 * R2
            if (nn_not_finite(m)) { slow = true; break; }                     // workgroup-uniform: m comes from LDS
        }
        if (wave_live) {
            // r04: the sweep keeps NO queue.  r01-r03 visited every tile with the running minimum (a hit test, a head in
            // registers, near-ties in LDS, an overflow path): ~13 VALU + two exec-mask branches per tile beside the 10 of the
            // v_min3 tree, and with 64 lanes some lane hits in almost every tile -- ~94 cycles per tile against the MFMA's 32.
            // Now a chunk of SCH tiles leaves only its surrogate minima in registers (MFMAs issued one group ahead of the trees
            // that read them: no wait states), then the threshold is known -- running minimum so far + tau: the final one can
            // only be lower, so the survivors are a superset -- and a branch-free pass marks the lane's survivors in a bit mask
            // (compare, shift-or); they are evaluated exactly, from a wave-wide work list (below).
            for (int c0 = 0; c0 < tn; c0 += SCH) {
                const int cn = min(SCH, tn - c0);                             // wave-uniform
                float mt[SCH];
                if (cn == SCH) {
                    const uint4 *fr = sfrag + c0 * 64 + lane;                 // tile u at fr[64 u]: immediate offsets
                    if constexpr (QW < 16) {
                        // fragments two tiles ahead of their MFMA, MFMAs one tile ahead of the tree that reads them.  (Smaller
                        // workgroups only: they serve small launches, where a wave has few neighbours to hide its LDS
                        // latency -- B = 12: 14.6 -> 13.3 us; the 16-wave form has four waves per SIMD, no use for it
                        // (B = 32: 20.6 vs 20.5) and no registers: it must fit 128.)
                        uint4 fa = fr[0], fb = fr[64];
                        f32x16 a = mfma(fa, bq);
                        fa = fr[2 * 64];
#pragma unroll
                        for (int u = 0; u < SCH; u += 2) {
                            const f32x16 b = mfma(fb, bq);
                            if (u + 3 < SCH) fb = fr[(u + 3) * 64];
                            __builtin_amdgcn_sched_barrier(0);
                            mt[u] = tile_min(a);
                            __builtin_amdgcn_sched_barrier(0);
                            if (u + 2 < SCH) {
                                a = mfma(fa, bq);
                                if (u + 4 < SCH) fa = fr[(u + 4) * 64];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            mt[u + 1] = tile_min(b);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        f32x16 a = mfma(fr[0], bq);
#pragma unroll
                        for (int u = 0; u < SCH; u += 2) {
                            const f32x16 b = mfma(fr[(u + 1) * 64], bq);
                            __builtin_amdgcn_sched_barrier(0);
                            mt[u] = tile_min(a);
                            __builtin_amdgcn_sched_barrier(0);
                            if (u + 2 < SCH) a = mfma(fr[(u + 2) * 64], bq);
                            __builtin_amdgcn_sched_barrier(0);
                            mt[u + 1] = tile_min(b);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                    // (a ragged last chunk repeats its last tile in the slots past it; their survivor bits are cleared below)
                    auto frag = [&](int u) { return sfrag[(c0 + min(u, cn - 1)) * 64 + lane]; };
                    f32x16 a = mfma(frag(0), bq);
#pragma unroll
                    for (int u = 0; u < SCH; u += 2) {
                        const f32x16 b = mfma(frag(u + 1), bq);
                        __builtin_amdgcn_sched_barrier(0);
                        mt[u] = tile_min(a);
                        __builtin_amdgcn_sched_barrier(0);
                        if (u + 2 < SCH) a = mfma(frag(u + 2), bq);
                        __builtin_amdgcn_sched_barrier(0);
                        mt[u + 1] = tile_min(b);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                float cm = fminf(fminf(mt[0], mt[1]), mt[2]);
#pragma unroll
                for (int u = 3; u + 1 < SCH; u += 2) cm = fminf(fminf(cm, mt[u]), mt[u + 1]);
                cm = fminf(cm, mt[SCH - 1]);
                smin = fminf(smin, cm);
                const float thr = fminf(smin, __shfl_xor(smin, 32)) + tau;
                // the lane's survivors as a bit mask (tile u at bit SCH - 1 - u): a compare and a shift-or per tile, no list
                unsigned smask = 0;
#pragma unroll
                for (int u = 0; u < SCH; ++u)          // smask = 2 smask + (mt[u] <= thr): a compare and an add-with-carry (the compiler's
                                                       // own form is compare, select, shift, or)
                    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(smask) : "v"(mt[u]), "v"(thr) : "vcc");
                // (a ragged last chunk repeated its last tile in slots cn .. SCH - 1: those bits name tiles that are not there)
                if (cn < SCH) smask &= ~0u << (SCH - cn);
                const int nsurv = __builtin_popcount(smask);
                int smax = nsurv;
                for (int d = 32; d > 0; d >>= 1) smax = max(smax, __shfl_xor(smax, d));
                smax = __builtin_amdgcn_readfirstlane(smax);
                // Exact evaluation.  A lane has 1-2 survivors per chunk on average but the wave's maximum is ~4, and a
                // round of the per-lane loop costs a whole exact_tile for every lane: instead the wave's survivors go into
                // ONE lane-major work list (exclusive scan of the counts), 64 items are evaluated per round by whichever
                // lane comes -- the owner's query arrives by shuffle -- and the owners merge their own results under the
                // (d, index) rule, which does not care about order.  (An adversarial chunk with more than WCAP items keeps
                // the per-lane loop.)
                int off = nsurv;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(off, d); off += lane >= d ? v : 0; }
                const int W = __builtin_amdgcn_readlane(off, 63);
                off -= nsurv;
                if (W <= WCAP) {
                    {
                        unsigned mk = smask;
                        for (int e = 0; e < smax; ++e) {
                            if (mk != 0) wl[off + e] = (unsigned short)((lane << 8) | (SCH - 1 - __builtin_ctz(mk)));
                            mk &= mk - 1;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int i0 = 0; i0 < W; i0 += 64) {
                        const int i = i0 + lane;
                        const unsigned item = wl[min(i, W - 1)];
                        const int sl = (int)(item >> 8), tl = c0 + (int)(item & 255u);
                        const float sx = __shfl(qx, sl), sy = __shfl(qy, sl), sz = __shfl(qz, sl);
                        float m; int k;
                        exact_tile_mk(spts, nc, t0 + tl, tl, sl >> 5, sx, sy, sz, m, k);
                        if (i < W) res[i] = make_uint2(__float_as_uint(m), (unsigned)k);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int e = 0; e < smax; ++e) {
                        if (e < nsurv) {
                            const uint2 r = res[off + e];
                            const float m = __uint_as_float(r.x);
                            const int k = (int)r.y;
                            const bool better = k != INT_MAX && (m < best || (m == best && k < bidx));
                            best = better ? m : best;
                            bidx = better ? k : bidx;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();                      // the results' space is the next chunk's
                } else {
                    unsigned mk = smask;
                    for (int e = 0; e < smax; ++e) {
                        const bool take = mk != 0;
                        const int tl = c0 + (take ? SCH - 1 - __builtin_ctz(mk) : 0);
                        mk &= mk - 1;
                        float b2 = best; int i2 = bidx;
                        exact_tile(spts, nc, t0 + tl, tl, h, qx, qy, qz, b2, i2);
                        if (take) { best = b2; bidx = i2; }
                    }
                }
            }
        }
    }
    if (slow && wave_live)          // both lane halves scan for their query: the merge below then finds them equal
        nn_reference_scan(cpts, nc, qx, qy, qz, best, bidx);
    if (!wave_live && !sums) return;
    // merge the two lane halves of each query
    const float od = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bidx, 32);
    if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
    // (the query index is made again from the thread index: carried from the top of the kernel it is the one value that does not
    // fit the 16-wave form's 128 registers)
    int tid_out = threadIdx.x;
    asm volatile("" : "+v"(tid_out));
    const int j_out = (blockIdx.x * QW + (tid_out >> 6)) * 32 + (tid_out & 31);
    if (!pairwise) {
        if (wave_live && h == 0 && j_out < nq) {
            A.dist[(size_t)bi * nq + j_out] = best;
            A.idx[(size_t)bi * nq + j_out] = bidx;
        }
        if (!sums) return;
    }
    // sum of this workgroup's distances in a fixed order -- butterfly over the 32 queries of a wave, waves in
    // ascending order -- so that the result does not depend on scheduling
    float sum = (wave_live && h == 0 && j_out < nq) ? best : 0.f;
    for (int d = 16; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    __syncthreads();                                   // s_r2 is free again (every wave has read tau's inputs long ago)
    if (lane == 0) s_r2[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        float t = s_r2[0];
#pragma unroll
        for (int w = 1; w < QW; ++w) t += s_r2[w];
        publish(t);
    }
}

// ---- the filter's surrogate, read back (tests/test_gpu_chamfer.py::test_filter_surrogate_error_bound) -------------------------
// One wave per (query tile, candidate tile) of ONE pair of clouds: the same centring, the same fragments and the same MFMA as
// nnm_kernel, and the 32 x 32 surrogates go to s[query][candidate] instead of into a minimum.  out3[0..2] = mu, out3[3] = R2 as
// nnm_kernel takes it (over all candidates and all queries; nnm_kernel's per-workgroup R2 is <= that).
__global__ __launch_bounds__(64) void nnm_surrogate_kernel(const float *__restrict__ q, int nq, const float *__restrict__ c, int nc,
                                                           float *__restrict__ s, float *__restrict__ out3) {
    __shared__ uint4 frag[64];
    const int lane = threadIdx.x, h = lane >> 5, i = lane & 31, qt = blockIdx.x, ct = blockIdx.y;
    float mux, muy, muz;
    {
        const float *src = c + (size_t)min(lane, nc - 1) * 3;
        mux = src[0]; muy = src[1]; muz = src[2];
        for (int d = 32; d > 0; d >>= 1) { mux += __shfl_xor(mux, d); muy += __shfl_xor(muy, d); muz += __shfl_xor(muz, d); }
        mux *= 0.015625f; muy *= 0.015625f; muz *= 0.015625f;
    }
    const int j = qt * 32 + i, p = ct * 32 + i;
    const float *qsrc = q + (size_t)min(j, nq - 1) * 3;
    const uint4 bq = query_fragment(qsrc[0] - mux, qsrc[1] - muy, qsrc[2] - muz, h);
    if (h == 0) {
        const bool live = p < nc;
        const float *src = c + (size_t)min(p, nc - 1) * 3;
        uint4 f0, f1;
        cand_fragment(live ? src[0] - mux : 0.f, live ? src[1] - muy : 0.f, live ? src[2] - muz : 0.f, live, f0, f1);
        frag[i] = f0; frag[32 + i] = f1;
    }
    __syncthreads();
    const f32x16 a = mfma(frag[lane], bq);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (j < nq && k < nc) s[(size_t)j * nc + k] = a[r];
    }
    if (qt == 0 && ct == 0) {          // R2 and mu, by one wave
        float r2 = 0.f;
#pragma clang loop vectorize(disable) interleave(disable)      // (no packed fp32 in a kernel that issues MFMAs: exact_tile_mk)
        for (int t = lane; t < nq + nc; t += 64) {
            const float *src = t < nq ? q + (size_t)t * 3 : c + (size_t)(t - nq) * 3;
            const float x = src[0] - mux, y = src[1] - muy, z = src[2] - muz;
            r2 = fmaxf(r2, (x * x + y * y) + z * z);
        }
        for (int d = 32; d > 0; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
        if (lane == 0) { out3[0] = mux; out3[1] = muy; out3[2] = muz; out3[3] = r2; }
    }
}

}  // namespace

// Test hook: the matrix-core filter's surrogate s(q, c) = |c - mu|^2 - 2 (q - mu).(c - mu) for every pair of ONE pair of clouds
// ((nq, 3) queries, (nc, 3) candidates), exactly as nnm_kernel forms it; s is (nq, nc), out4 = {mu_x, mu_y, mu_z, R2}.
extern "C" int dpf_debug_nn_surrogate(int nq, const float *q, int nc, const float *c, float *s, float *out4, dpf_stream_t stream) {
    if (nq <= 0 || nc <= 0 || !q || !c || !s || !out4) return DPF_EINVAL;
    hipLaunchKernelGGL(nnm_surrogate_kernel, dim3(tiles_of(nq), tiles_of(nc)), dim3(64), 0, (hipStream_t)stream, q, nq, c, nc, s, out4);
    return (int)hipGetLastError();
}

extern "C" size_t dpf_nndistance_mfma_workspace_bytes(int b, int n, int m) {
    (void)b; (void)n; (void)m;
    return 0;            // the fragments are built inside the kernel since r01; the argument is kept for the ABI
}

static long nnm_workgroups(int b, int n, int m, int qw) {
    return (long)b * ((n + qw * 32 - 1) / (qw * 32) + (m + qw * 32 - 1) / (qw * 32));
}

template <int QW>
static int launch_nnm_qw(const MArgs &ma, int b, int nmax, hipStream_t s) {
    // (the grid's x extent is the stride of the partial sums when ma.part is set)
    const int lds = nnm_lds_bytes(QW);
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)nnm_kernel<QW>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(nnm_kernel<QW>, dim3((nmax + QW * 32 - 1) / (QW * 32), b, 2), dim3(QW * 64), lds, s, ma);
    return (int)hipGetLastError();
}

// 16 waves (512 queries) per workgroup share one build of the candidates' fragments; when that leaves fewer than 128
// workgroups (small batches of big clouds, e.g. B = 2, N = 8192 per GPU in cfg-5) 8-wave workgroups fill twice the CUs
static int launch_nnm(int b, int n, const float *xyz, long xyz_stride, int m, const float *xyz2, long xyz2_stride,
                      float *result, int *result_i, float *result2, int *result2_i, hipStream_t s, float *part = nullptr,
                      bool force16 = false, unsigned *ticket = nullptr, float *cd = nullptr) {
    MArgs ma;
    ma.pn2 = 0; ma.part = part; ma.ticket = ticket; ma.cd = cd;
    ma.d[0] = MDir{xyz, xyz2, result, result_i, n, m, xyz_stride, xyz2_stride};       // nndistance.cu:126
    ma.d[1] = MDir{xyz2, xyz, result2, result2_i, m, n, xyz2_stride, xyz_stride};     // nndistance.cu:127
    const int nmax = n > m ? n : m;
    static const int qw_env = getenv("DPF_NNM_QW") ? atoi(getenv("DPF_NNM_QW")) : 0;      // experiments: 4 | 8 | 16
    if (force16 || qw_env == 16) return launch_nnm_qw<16>(ma, b, nmax, s);
    if (qw_env == 8) return launch_nnm_qw<8>(ma, b, nmax, s);
    if (qw_env == 4) return launch_nnm_qw<4>(ma, b, nmax, s);
    if (nnm_workgroups(b, n, m, 16) >= 128) return launch_nnm_qw<16>(ma, b, nmax, s);
    return nnm_workgroups(b, n, m, 8) >= 128 ? launch_nnm_qw<8>(ma, b, nmax, s) : launch_nnm_qw<4>(ma, b, nmax, s);
}

// enough pairs to amortise building the fragments and enough workgroups to fill the chip (r01, tools/nn_impl_sweep.py:
// 25 vs 52 us at B=32, n=m=2048; 76 vs 190 us at B=8, n=m=8192; 50 vs 64 us at B=2, n=m=8192); small clouds and small
// batches are launch-bound either way and few workgroups leave the matrix cores idle
// r04, after the filter's bookkeeping was rebuilt (see nnm_kernel): it also wins for mid-sized batches of clouds whose
// fragments fit one pass -- B = 6 / 8 / 10 clouds of 2048: 13.4 / 12.0 / 12.1 us against the scans' 15.2 / 15.3 / 24.4; 16 clouds
// of 1024: 8.2 vs 10.0 -- but not where the LDS-staged scan serves (one workgroup per CU or fewer: B = 4: 10.4 vs 13.3) and not
// for clouds a little over one pass (B = 4, N = 2500: 22.8 vs 17.3).
bool nn_small_serves(int b, int n, int m);      // chamfer.hip
static bool nnm_pays(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0 || b > 65535 || n > 65535 * 32 || m > 65535 * 32) return false;
    if (nnm_workgroups(b, n, m, 8) < 64) return false;                // B=1, n=m=8192: 48 vs 57 us
    const double pairs = 2.0 * (double)b * (double)n * (double)m;
    if (pairs >= 1.0e8) return true;
    return pairs >= 3.0e7 && (n > m ? n : m) <= CT * 32 && !nn_small_serves(b, n, m);
}

extern "C" int dpf_nndistance_mfma(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                   int *result_i, float *result2, int *result2_i, void *workspace,
                                   size_t workspace_bytes, dpf_stream_t stream) {
    (void)workspace; (void)workspace_bytes;
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    if (b > 65535 || n > 65535 * 32 || m > 65535 * 32 || (n < 32 && m < 32))
        return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    return launch_nnm(b, n, xyz, (long)n * 3, m, xyz2, (long)m * 3, result, result_i, result2, result2_i, (hipStream_t)stream);
}

// Same contract and the same bits as dpf_nndistance; the matrix-core filtered kernel where it measured faster, the
// VALU scan otherwise.
extern "C" int dpf_nndistance_auto(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                                   float *result2, int *result2_i, dpf_stream_t stream) {
    if (xyz && xyz2 && result && result_i && result2 && result2_i && nnm_pays(b, n, m))
        return launch_nnm(b, n, xyz, (long)n * 3, m, xyz2, (long)m * 3, result, result_i, result2, result2_i, (hipStream_t)stream);
    return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
}

// dpf_nndistance_strided's contract and bits (explicit per-cloud strides, 0 = broadcast: one row of pairwise_CD per
// launch) with the same choice of kernel.
extern "C" int dpf_nndistance_strided_auto(int b, int n, const float *xyz, long xyz_stride, int m, const float *xyz2,
                                           long xyz2_stride, float *result, int *result_i, float *result2, int *result2_i,
                                           dpf_stream_t stream) {
    if (xyz && xyz2 && result && result_i && result2 && result2_i && xyz_stride >= 0 && xyz2_stride >= 0 && nnm_pays(b, n, m))
        return launch_nnm(b, n, xyz, xyz_stride, m, xyz2, xyz2_stride, result, result_i, result2, result2_i, (hipStream_t)stream);
    return dpf_nndistance_strided(b, n, xyz, xyz_stride, m, xyz2, xyz2_stride, result, result_i, result2, result2_i, stream);
}

// The whole (N1, N2) Chamfer-distance matrix of lib/networks/utils.py:90-117 (pairwise_CD) in ONE launch (+ one tiny finish):
// cds[i, j] = mean_k min_l |a_ik - b_jl|^2 + mean_l min_k |a_ik - b_jl|^2 for clouds1 (N1, n, 3) and clouds2 (N2, m, 3).
// The reference expands cloud i against a batch of clouds2, copies it, and calls nn_distance once per i (utils.py:104-107);
// here grid = (query tiles, j, 2 i + direction), every workgroup reads its two clouds in place, the per-point distances
// are summed in the kernel and never written.  workspace: dpf_pairwise_cd_workspace_bytes(N1, N2, n, m) bytes.
// Rows [i0, i1) only: what one rank of a row-sharded evaluation computes (cds still has N2 columns).
// 512-query workgroups; 256-query ones for clouds of <= 256 points (half of a 16-wave workgroup would idle)
static int pairwise_qw(int nmax) { return nmax <= 256 ? 8 : 16; }
extern "C" size_t dpf_pairwise_cd_workspace_bytes(int n1, int n2, int n, int m) {
    const int nmax = n > m ? n : m, per = pairwise_qw(nmax) * 32;
    return (size_t)n1 * n2 * (2 * ((nmax + per - 1) / per) + 1) * sizeof(float);      // sums + one ticket per pair
}
extern "C" int dpf_pairwise_cd(int n1, int n2, int n, int m, const float *clouds1, const float *clouds2, float *cds,
                               void *workspace, size_t workspace_bytes, dpf_stream_t stream) {
    if (n1 < 0 || n2 < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    if (!clouds1 || !clouds2 || !cds || !workspace) return DPF_EINVAL;
    if (workspace_bytes < dpf_pairwise_cd_workspace_bytes(n1, n2, n, m)) return DPF_EINVAL;
    if (n2 > 65535 || n1 > 32767 || n < 32 || m < 32 || n > 65535 * 32 || m > 65535 * 32) return DPF_ENOSUP;
    MArgs ma;
    ma.d[0] = MDir{clouds1, clouds2, nullptr, nullptr, n, m, (long)n * 3, (long)m * 3};
    ma.d[1] = MDir{clouds2, clouds1, nullptr, nullptr, m, n, (long)m * 3, (long)n * 3};
    const int nmax = n > m ? n : m, qw = pairwise_qw(nmax), nwg = (nmax + qw * 32 - 1) / (qw * 32);
    const long npairs = (long)n1 * n2;
    hipStream_t s = (hipStream_t)stream;
    ma.pn2 = n2; ma.ticket = (unsigned *)workspace; ma.part = (float *)workspace + npairs; ma.cd = cds;
    // the tickets (first n1 * n2 words) start at zero; the last arriver of every pair resets its own
    if (hipError_t e = dpf_zero_async(workspace, (size_t)npairs * sizeof(unsigned), s); e != hipSuccess) return (int)e;
    if (qw == 16) {
        const int lds = nnm_lds_bytes(16);
        static LdsLimit limit;
        if (hipError_t e = limit.ensure((const void *)nnm_kernel<16>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nnm_kernel<16>, dim3(nwg, n2, 2 * n1), dim3(16 * 64), lds, s, ma);
    } else {
        const int lds = nnm_lds_bytes(8);
        static LdsLimit limit;
        if (hipError_t e = limit.ensure((const void *)nnm_kernel<8>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nnm_kernel<8>, dim3(nwg, n2, 2 * n1), dim3(8 * 64), lds, s, ma);
    }
    return (int)hipGetLastError();
}

// nn_distance + the per-cloud reduction of its callers in one go (lib/networks/evaluating.py:110-113:
// `dl, dr = distChamferCUDA(...); cd = (dl.mean(1) + dr.mean(1))`): the outputs of dpf_nndistance (same bits) and
// cd[b] = mean(result[b]) + mean(result2[b]).  Where the matrix-core kernel serves the problem its workgroups also emit
// fixed-order sums of their distances and a tiny finish kernel adds them (workspace: dpf_nndistance_cd_workspace_bytes);
// otherwise dpf_nndistance + dpf_chamfer_reduce.  The two reductions associate differently (last-bit differences in cd).
extern "C" size_t dpf_nndistance_cd_workspace_bytes(int b, int n, int m) {
    const int nmax = n > m ? n : m;
    // sums (one per 64-query workgroup of nn_small_kernel, the finest tiling) + one ticket per cloud
    return (size_t)(b > 0 ? b : 0) * (2 * ((nmax + 63) / 64) + 1) * sizeof(float) + 16;
}
int nn_small_cd(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i, float *result2,
                int *result2_i, float *cd, void *workspace, int tickets_are_zero, hipStream_t s);      // chamfer.hip
extern "C" int dpf_nndistance_cd(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                                 float *result2, int *result2_i, float *cd, void *workspace, size_t workspace_bytes,
                                 int tickets_are_zero, dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i || !cd) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (workspace && workspace_bytes >= dpf_nndistance_cd_workspace_bytes(b, n, m) && nnm_pays(b, n, m)) {
        // tickets (first b words): zero on entry -- `tickets_are_zero` = 0 makes this call clear them first -- and zero again
        // on exit, so a caller that keeps the workspace pays the memset once
        if (!tickets_are_zero)
            if (hipError_t e = dpf_zero_async(workspace, (size_t)b * sizeof(unsigned), s); e != hipSuccess) return (int)e;
        // (the workgroup size is launch_nnm's choice: the scratch has room for the finest tiling's sums)
        return launch_nnm(b, n, xyz, (long)n * 3, m, xyz2, (long)m * 3, result, result_i, result2, result2_i, s,
                          (float *)workspace + b, false, (unsigned *)workspace, cd);
    }
    if (workspace && workspace_bytes >= dpf_nndistance_cd_workspace_bytes(b, n, m) && !nnm_pays(b, n, m)) {
        // a rank's handful of clouds: the LDS-staged scan finishes the reduction the same way (chamfer.hip nn_small_kernel)
        const int rc = nn_small_cd(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, cd, workspace, tickets_are_zero, s);
        if (rc != DPF_ENOSUP) return rc;
    }
    int rc = dpf_nndistance_auto(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    if (rc) return rc;
    return dpf_chamfer_reduce(b, n, m, result, result2, cd, stream);
}
