// Exact Chamfer nearest neighbour with a quantile 3-D grid, for gfx950 (MI355X).
//
// Contract = nn_kernel in chamfer.hip: bit-exact d = (dx*dx + dy*dy) + dz*dz (no FMA) and the
// LOWEST original index among exact ties, i.e. the result of the reference's strict '<' scan in
// ascending k (lib/metrics/pytorch_structural_losses/src/nndistance.cu:16-119) -- but only the
// candidates that can still win are evaluated:
//
//  1. grid_build_kernel (one workgroup per cloud, both clouds of a pair in one launch): per axis
//     G-1 cell boundaries at the k/G quantiles of the coordinates (from a 1024-bin histogram), so
//     that a Gaussian blob or a surface fills the cells as evenly as a uniform cube;
//     cell(x) = #{k : x >= b_k} -- pure comparisons, the same rule for points and queries;
//     counting sort of the points by cell in LDS.  Output: points in cell order as float4
//     (x, y, z, original index), the cell start table, the boundaries.
//  2. grid_query_kernel: a workgroup stages the whole candidate cloud (sorted points, start table,
//     boundaries: 37 KB for n = 2048, 148 KB for n = 8192) in LDS with direct global->LDS loads; each
//     LANE then walks the cells within r of its own query's cell -- (2r+1)^2 rows, a row being one
//     contiguous range of the sorted points -- with the exact fp32 formula and the (d, original
//     index) lexicographic rule.  A lane is DONE when its best distance is strictly below a bound
//     that holds for every unscanned point: for each face b of the scanned box with cells behind it,
//     e_axis = fl(b - q) on the face's axis and, on the other two axes, the distance from q to the
//     candidates' bounding box (0 if inside); LB_face = fl(fl(ex*ex + ey*ey) + ez*ez) in the
//     association order of the distance itself.  An unscanned point c behind that face has
//     |fl(c - q)| >= |e| on every axis (rounding is monotone), hence d(c) >= LB_face > best: it can
//     neither win nor tie.  The few lanes that are not done after the r = 1 box (their nearest
//     neighbour is further than a cell away: sparse tails) are resolved one at a time by the WHOLE
//     wave: all 64 lanes scan the complete candidate table for that query and reduce with the same
//     lexicographic rule.  No tolerance anywhere: exact for any finite input; the worst case
//     (every query a straggler) costs a full scan per query, never a wrong answer.
//
// Status (r01, MI355X, measured with tools/nn_grid_dbg.py; brute force = nn_kernel in chamfer.hip):
//   B=32 N=2048  uniform cube  41 us (13 build + 26 query) vs 53 brute;  torus surface 145 vs 52;
//                Gaussian blob 124 vs 55;  bench clouds (flow output vs cube) 266 vs 51
//   B=16 N=8192  uniform cube 104 vs 369;  surface 968 vs 379;  Gaussian 1093 vs 363
// i.e. it wins only when the cells are isotropic -- on surfaces and tails the marginal-quantile cells are
// elongated, a third of the lanes become stragglers and the per-query full scans dominate.  Opt-in
// (DPF_CHAMFER_IMPL=grid); the brute-force kernel at ~75 % of its VALU bound stays the default.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "dpf_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int GMAX = 16;                 // cells per axis
constexpr int HB = 1024;                 // histogram bins per axis for the quantiles
constexpr int BUILD_THREADS = 1024;
constexpr int QWAVES = 8;
constexpr int LDS_LIMIT = 160 * 1024;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__host__ __device__ inline int pad_pts(int n) { return (n + 63) / 64 * 64 + 64; }     // whole 1 KiB lines + sentinels
__host__ __device__ inline int grid_dim_for(int n) {
    int g = 2;
    while (g < GMAX && (g + 1) * (g + 1) * (g + 1) * 2 <= n) ++g;
    return g;
}
// per-cloud table: start[G^3 + 1] ints, then bounds[3][GMAX + 1] floats (b_0 = -inf, b_G = +inf), then the
// bounding box min[3] max[3]; padded to 1 KiB
__host__ __device__ inline int table_bytes(int G) { return ((G * G * G + 1) * 4 + 3 * (GMAX + 1) * 4 + 8 * 4 + 1023) / 1024 * 1024; }
__host__ __device__ inline int bounds_off(int G) { return (G * G * G + 1) * 4; }

struct GridSet {
    const float *xyz;    // (B, n, 3)
    float4 *spts;        // (B, npad)  sorted (x, y, z, original index as bits); tail = (+inf, 0, 0, INT_MAX)
    uint8_t *table;      // (B, table_bytes)
    int n, npad, G;
};
struct BuildArgs { GridSet s[2]; };

__device__ __forceinline__ int cell_of(float x, const float *b, int G) {     // #{k in 1..G-1 : x >= b[k]}
    int c = 0;
    for (int k = 1; k < G; ++k) c += x >= b[k] ? 1 : 0;
    return c;
}

constexpr int PPT = 8;                   // points per thread held in registers by the build kernel (n <= 8192)

__global__ __launch_bounds__(BUILD_THREADS) void grid_build_kernel(BuildArgs args) {
    __shared__ float red[6][BUILD_THREADS / 64];
    __shared__ float bx[8];
    __shared__ float bnd[3][GMAX + 1];
    __shared__ int hist[3][HB];
    __shared__ int cnt[GMAX * GMAX * GMAX + 1];
    __shared__ int wsum[BUILD_THREADS / 64];
    const GridSet S = args.s[blockIdx.y];
    const int bi = blockIdx.x, n = S.n, G = S.G, NC = G * G * G;
    const float *src = S.xyz + (size_t)bi * n * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- the cloud is read once; every later phase works on registers and LDS
    float px[PPT], py[PPT], pz[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = threadIdx.x + u * BUILD_THREADS;
        const bool ok = i < n;
        px[u] = ok ? src[i * 3] : 0.f; py[u] = ok ? src[i * 3 + 1] : 0.f; pz[u] = ok ? src[i * 3 + 2] : 0.f;
    }
    // ---- bounding box
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < PPT; ++u)
        if (threadIdx.x + u * BUILD_THREADS < n) {
            mn[0] = fminf(mn[0], px[u]); mx[0] = fmaxf(mx[0], px[u]);
            mn[1] = fminf(mn[1], py[u]); mx[1] = fmaxf(mx[1], py[u]);
            mn[2] = fminf(mn[2], pz[u]); mx[2] = fmaxf(mx[2], pz[u]);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
    if (lane == 0)
        for (int c = 0; c < 3; ++c) { red[c][wave] = mn[c]; red[3 + c][wave] = mx[c]; }
    for (int i = threadIdx.x; i <= NC; i += BUILD_THREADS) cnt[i] = 0;
    for (int i = threadIdx.x; i < 3 * HB; i += BUILD_THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    if (threadIdx.x < 3) {
        float a = red[threadIdx.x][0], b = red[3 + threadIdx.x][0];
        for (int w = 1; w < BUILD_THREADS / 64; ++w) { a = fminf(a, red[threadIdx.x][w]); b = fmaxf(b, red[3 + threadIdx.x][w]); }
        const float ext = b - a;
        bx[threadIdx.x] = a;
        bx[3 + threadIdx.x] = ext > 0.f && ext < INFINITY ? ext : 0.f;
        red[threadIdx.x][0] = a; red[3 + threadIdx.x][0] = b;           // min / max for the table
    }
    __syncthreads();
    // ---- per-axis histogram -> quantile boundaries
    float sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) sc[c] = bx[3 + c] > 0.f ? (float)HB / bx[3 + c] : 0.f;
#pragma unroll
    for (int u = 0; u < PPT; ++u)
        if (threadIdx.x + u * BUILD_THREADS < n) {
            const float v[3] = {px[u], py[u], pz[u]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int b = (int)((v[c] - bx[c]) * sc[c]);
                b = b < 0 ? 0 : (b >= HB ? HB - 1 : b);
                atomicAdd(&hist[c][b], 1);
            }
        }
    __syncthreads();
    if (wave < 3) {                                   // inclusive scan of one axis per wave: 64 lanes x 16 bins
        int *hc = hist[wave];
        int s = 0;
        for (int k = 0; k < HB / 64; ++k) s += hc[lane * (HB / 64) + k];
        int inc = s;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        int run = inc - s;
        for (int k = 0; k < HB / 64; ++k) { run += hc[lane * (HB / 64) + k]; hc[lane * (HB / 64) + k] = run; }
    }
    __syncthreads();
    if (threadIdx.x < 3 * (GMAX + 1)) {
        const int c = threadIdx.x / (GMAX + 1), k = threadIdx.x % (GMAX + 1);
        float b;
        if (k == 0) b = -INFINITY;
        else if (k >= G) b = INFINITY;
        else {
            const int target = (int)(((long)k * n + G - 1) / G);
            int lo = 0, hi = HB - 1;                  // smallest bin whose cumulative count reaches the target
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (hist[c][mid] >= target) hi = mid; else lo = mid + 1; }
            b = bx[c] + (float)(lo + 1) * (bx[3 + c] / (float)HB);
        }
        bnd[c][k] = b;
    }
    __syncthreads();
    // ---- cell histogram, exclusive scan, scatter (counting sort by cell)
    int id[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u)
        if (threadIdx.x + u * BUILD_THREADS < n) {
            id[u] = (cell_of(pz[u], bnd[2], G) * G + cell_of(py[u], bnd[1], G)) * G + cell_of(px[u], bnd[0], G);
            atomicAdd(&cnt[id[u]], 1);
        }
    __syncthreads();
    {   // block-wide exclusive scan of cnt[0..NC): each thread owns a strip of consecutive cells
        const int per = (NC + BUILD_THREADS - 1) / BUILD_THREADS;
        const int b0 = threadIdx.x * per;
        int s = 0;
        for (int k = 0; k < per; ++k) if (b0 + k < NC) s += cnt[b0 + k];
        int inc = s;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + inc - s;
        for (int k = 0; k < per; ++k)
            if (b0 + k < NC) { const int c = cnt[b0 + k]; cnt[b0 + k] = run; run += c; }
        if (threadIdx.x == 0) cnt[NC] = n;
    }
    __syncthreads();
    uint8_t *tab = S.table + (size_t)bi * table_bytes(G);
    int *start = (int *)tab;
    for (int i = threadIdx.x; i <= NC; i += BUILD_THREADS) start[i] = cnt[i];
    float *bo = (float *)(tab + bounds_off(G));
    if (threadIdx.x < 3 * (GMAX + 1)) bo[threadIdx.x] = (&bnd[0][0])[threadIdx.x];
    if (threadIdx.x < 6) bo[3 * (GMAX + 1) + threadIdx.x] = red[threadIdx.x][0];
    __syncthreads();
    float4 *dst = S.spts + (size_t)bi * S.npad;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = threadIdx.x + u * BUILD_THREADS;
        if (i < n) {
            const int pos = atomicAdd(&cnt[id[u]], 1);
            dst[pos] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
        }
    }
    for (int i = n + threadIdx.x; i < S.npad; i += BUILD_THREADS) dst[i] = make_float4(INFINITY, 0.f, 0.f, __int_as_float(INT_MAX));
}

struct QDir {
    const float4 *q;     // sorted queries (B, nqpad)
    const float4 *c;     // sorted candidates (B, ncpad)
    const uint8_t *ctab; // (B, table_bytes(G))
    float *dist;         // (B, nq) in ORIGINAL query order
    int *idx;
    int nq, nqpad, nc, ncpad, G;
};
struct QArgs { QDir d[2]; };

__device__ __forceinline__ void consider(float4 p, bool valid, float qx, float qy, float qz, float &best, int &bidx) {
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    const int oi = __float_as_int(p.w);
    const bool better = valid && (d < best || (d == best && oi < bidx));
    best = better ? d : best;
    bidx = better ? oi : bidx;
}

__global__ __launch_bounds__(QWAVES * 64) void grid_query_kernel(QArgs args) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const QDir A = args.d[blockIdx.z];
    const int bi = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x * QWAVES * 64 >= A.nq) return;                         // whole workgroup out of range (uniform)
    const int G = A.G;
    // ---- stage the candidate cloud: sorted points | start table + boundaries
    const int pbytes = A.ncpad * 16, tbytes = table_bytes(G);
    {
        const uint8_t *psrc = (const uint8_t *)(A.c + (size_t)bi * A.ncpad);
        const uint8_t *tsrc = A.ctab + (size_t)bi * tbytes;
        const int np = pbytes / 1024, nt = tbytes / 1024;
        for (int ch = wave; ch < np + nt; ch += QWAVES) {
            const uint8_t *src = (ch < np ? psrc + ch * 1024 : tsrc + (ch - np) * 1024) + lane * 16;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + ch * 1024), 16, 0, 0);
        }
    }
    const float4 *P = (const float4 *)smem;
    const int *S = (const int *)(smem + pbytes);
    const float *bnd = (const float *)(smem + pbytes + bounds_off(G));
    const int j = (blockIdx.x * QWAVES + wave) * 64 + lane;
    const bool live = j < A.nq;
    const float4 qv = A.q[(size_t)bi * A.nqpad + (live ? j : A.nq - 1)];
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float *bX = bnd, *bY = bnd + (GMAX + 1), *bZ = bnd + 2 * (GMAX + 1), *bb = bnd + 3 * (GMAX + 1);
    const int cx = cell_of(qx, bX, G), cy = cell_of(qy, bY, G), cz = cell_of(qz, bZ, G);
    float best = INFINITY;
    int bidx = INT_MAX;
    // ---- pass 1: every lane walks the cells within one of its query's cell
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G - 1), y0 = max(cy - 1, 0), y1 = min(cy + 1, G - 1);
    const int z0 = max(cz - 1, 0), z1 = min(cz + 1, G - 1);
    {
        int yy = y0, zz = z0;
        bool rows_left = live;
        while (__builtin_amdgcn_ballot_w64(rows_left) != 0) {
            const int row = (zz * G + yy) * G;
            int k = rows_left ? S[row + x0] : 0;
            const int k1 = rows_left ? S[row + x1 + 1] : 0;
            while (__builtin_amdgcn_ballot_w64(k < k1) != 0) {           // 4 candidates per trip, per-lane ranges
                const float4 p0 = P[k], p1 = P[k + 1], p2 = P[k + 2], p3 = P[k + 3];     // reads past k1 stay inside the padded table
                consider(p0, k < k1, qx, qy, qz, best, bidx);
                consider(p1, k + 1 < k1, qx, qy, qz, best, bidx);
                consider(p2, k + 2 < k1, qx, qy, qz, best, bidx);
                consider(p3, k + 3 < k1, qx, qy, qz, best, bidx);
                k = k < k1 ? k + 4 : k;
            }
            if (rows_left) {                                              // next (y, z) row of this lane's box
                if (++yy > y1) { yy = y0; ++zz; }
                rows_left = zz <= z1;
            }
        }
    }
    // ---- strict lower bound of every unscanned point: per face with cells behind it, the face distance on its
    // axis and the distance to the candidates' bounding box on the other two, combined like the distance itself
    bool done;
    {
        const float ox = fmaxf(fmaxf(bb[0] - qx, qx - bb[3]), 0.f), oy = fmaxf(fmaxf(bb[1] - qy, qy - bb[4]), 0.f),
                    oz = fmaxf(fmaxf(bb[2] - qz, qz - bb[5]), 0.f);
        float lb = INFINITY;
        float t;
        if (x0 > 0) { t = bX[x0] - qx; lb = fminf(lb, (t * t + oy * oy) + oz * oz); }
        if (x1 < G - 1) { t = bX[x1 + 1] - qx; lb = fminf(lb, (t * t + oy * oy) + oz * oz); }
        if (y0 > 0) { t = bY[y0] - qy; lb = fminf(lb, (ox * ox + t * t) + oz * oz); }
        if (y1 < G - 1) { t = bY[y1 + 1] - qy; lb = fminf(lb, (ox * ox + t * t) + oz * oz); }
        if (z0 > 0) { t = bZ[z0] - qz; lb = fminf(lb, (ox * ox + oy * oy) + t * t); }
        if (z1 < G - 1) { t = bZ[z1 + 1] - qz; lb = fminf(lb, (ox * ox + oy * oy) + t * t); }
        const bool whole = x0 == 0 && x1 == G - 1 && y0 == 0 && y1 == G - 1 && z0 == 0 && z1 == G - 1;
        done = !live || whole || best < lb;
    }
    // ---- stragglers: one query at a time, the whole wave scans the complete candidate table
    unsigned long long todo = __builtin_amdgcn_ballot_w64(!done);
    while (todo != 0) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const float sx = __shfl(qx, l), sy = __shfl(qy, l), sz = __shfl(qz, l);
        float b2 = INFINITY;
        int i2 = INT_MAX;
        for (int k = lane; k < A.nc; k += 64) consider(P[k], true, sx, sy, sz, b2, i2);
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(b2, o);
            const int oi = __shfl_xor(i2, o);
            const bool better = ob < b2 || (ob == b2 && oi < i2);
            b2 = better ? ob : b2;
            i2 = better ? oi : i2;
        }
        if (lane == l) { best = b2; bidx = i2; }
    }
    if (live) {
        const int o = __float_as_int(qv.w);
        A.dist[(size_t)bi * A.nq + o] = best;
        A.idx[(size_t)bi * A.nq + o] = bidx;
    }
}

struct WsLayout { size_t spts[2], table[2], total; };
WsLayout layout(int b, int n, int m) {
    WsLayout w;
    const int nn[2] = {n, m};
    size_t off = 0;
    for (int s = 0; s < 2; ++s) {
        const int G = grid_dim_for(nn[s]);
        w.spts[s] = off; off += (size_t)b * pad_pts(nn[s]) * sizeof(float4);
        w.table[s] = off; off += (size_t)b * table_bytes(G);
    }
    w.total = off;
    return w;
}
int query_lds(int nc) { return pad_pts(nc) * 16 + table_bytes(grid_dim_for(nc)); }

}  // namespace

extern "C" size_t dpf_nndistance_grid_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return layout(b, n, m).total;
}

extern "C" int dpf_nndistance_grid(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                                   float *result2, int *result2_i, void *workspace, size_t workspace_bytes,
                                   dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    // too small to be worth two extra passes, a cloud that does not fit in LDS, or no workspace: the brute-force kernel
    const int lds = query_lds(n) > query_lds(m) ? query_lds(n) : query_lds(m);
    if (n < 256 || m < 256 || n > PPT * BUILD_THREADS || m > PPT * BUILD_THREADS || b > 65535 || lds > LDS_LIMIT || !workspace ||
        workspace_bytes < dpf_nndistance_grid_workspace_bytes(b, n, m))
        return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    hipStream_t s = (hipStream_t)stream;
    const WsLayout w = layout(b, n, m);
    uint8_t *ws = (uint8_t *)workspace;
    BuildArgs ba;
    const float *src[2] = {xyz, xyz2};
    const int nn[2] = {n, m};
    for (int k = 0; k < 2; ++k)
        ba.s[k] = GridSet{src[k], (float4 *)(ws + w.spts[k]), ws + w.table[k], nn[k], pad_pts(nn[k]), grid_dim_for(nn[k])};
    hipLaunchKernelGGL(grid_build_kernel, dim3(b, 2), dim3(BUILD_THREADS), 0, s, ba);
    QArgs qa;
    // direction 0: queries = cloud 1 (xyz), candidates = cloud 2   (nndistance.cu:126); direction 1 the reverse (:127)
    qa.d[0] = QDir{ba.s[0].spts, ba.s[1].spts, ba.s[1].table, result, result_i, n, ba.s[0].npad, m, ba.s[1].npad, ba.s[1].G};
    qa.d[1] = QDir{ba.s[1].spts, ba.s[0].spts, ba.s[0].table, result2, result2_i, m, ba.s[1].npad, n, ba.s[0].npad, ba.s[0].G};
    static int attr_lds = 0;
    if (lds > 65536 && lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void *)grid_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_lds = lds;
    }
    const int nmax = n > m ? n : m;
    hipLaunchKernelGGL(grid_query_kernel, dim3((nmax + QWAVES * 64 - 1) / (QWAVES * 64), b, 2), dim3(QWAVES * 64), lds, s, qa);
    return (int)hipGetLastError();
}
