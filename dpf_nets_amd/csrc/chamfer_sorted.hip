// Exact Chamfer nearest neighbour with 1-D pruning, for gfx950 (MI355X).
//
// Same contract as nn_kernel in chamfer.hip -- bit-exact distances
// d = (dx*dx + dy*dy) + dz*dz (no FMA), lowest original index among exact ties,
// i.e. the result of the reference's strict '<' scan in ascending k
// (lib/metrics/pytorch_structural_losses/src/nndistance.cu:16-119) -- but it
// evaluates only the candidates that can still win:
//
//   1. sort_kernel: every cloud is sorted by x once (bitonic sort in LDS, one
//      workgroup per cloud), emitting the points in sorted order and their
//      original indices.  Each cloud serves as candidate set for one direction
//      and as (x-coherent) query set for the other.
//   2. nnq_kernel: a wave owns 64 consecutive sorted queries, finds the sorted
//      candidate position of its median query, and expands right and left in
//      chunks of 8 wave-uniform candidates (scalar loads -> SGPR operands of
//      plain fp32 VALU ops).  A direction stops when, for every lane,
//      fl((x_k - q_x)^2) > best: in floating point d >= fl(dx*dx) (adding
//      non-negative terms is monotone under rounding) and fl(dx*dx) is monotone
//      along the sorted order, so no later candidate can be closer OR tie.
//      The chunk loop only tracks the running minimum; a chunk in which some
//      lane reaches d <= best is re-evaluated candidate by candidate with the
//      (d, original index) lexicographic rule.
//
// For N = 2048 points in a cube this evaluates ~10 % of the n*m pairs.  Worst
// case (all x equal) degenerates to the full scan and stays exact.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "dpf_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int CH = 8;
constexpr int MAXN = 8192;        // bitonic sort capacity (LDS: 8 B per element)

__device__ __forceinline__ uint32_t ordered_key(float f) {   // total order on floats as uint32
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__host__ __device__ inline int pad8(int n) { return (n + CH - 1) / CH * CH; }

// workspace per cloud set: sorted AoS points [B][npad][3] floats, original index [B][npad] ints
struct SortSet {
    const float *xyz;   // (B, n, 3)
    float *spts;        // (B, npad, 3)
    int *sidx;          // (B, npad)
    int n, npad;
};
struct SortArgs { SortSet s[2]; };

__global__ __launch_bounds__(1024) void sort_kernel(SortArgs args) {
    extern __shared__ uint32_t ssm[];
    const SortSet S = args.s[blockIdx.y];
    const int bi = blockIdx.x, n = S.n;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    uint32_t *key = ssm;
    int *val = (int *)(ssm + np2);
    const float *src = S.xyz + (size_t)bi * n * 3;
    for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        key[i] = i < n ? ordered_key(src[i * 3]) : 0xFFFFFFFFu;
        val[i] = i < n ? i : INT_MAX;
    }
    __syncthreads();
    // bitonic sort on (key, original index): ties in x keep ascending original index
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // index with bit j clear
                const int p = i | j;
                const bool up = (i & k) == 0;
                const uint32_t ka = key[i], kb = key[p];
                const int va = val[i], vb = val[p];
                const bool gt = ka > kb || (ka == kb && va > vb);
                if (gt == up) { key[i] = kb; key[p] = ka; val[i] = vb; val[p] = va; }
            }
            __syncthreads();
        }
    }
    float *dp = S.spts + (size_t)bi * S.npad * 3;
    int *di = S.sidx + (size_t)bi * S.npad;
    for (int i = threadIdx.x; i < S.npad; i += blockDim.x) {
        if (i < n) {
            const int o = val[i];
            dp[i * 3 + 0] = src[o * 3 + 0]; dp[i * 3 + 1] = src[o * 3 + 1]; dp[i * 3 + 2] = src[o * 3 + 2];
            di[i] = o;
        } else {   // sentinel: +inf distance, never wins a tie
            dp[i * 3 + 0] = __builtin_inff(); dp[i * 3 + 1] = 0.f; dp[i * 3 + 2] = 0.f;
            di[i] = INT_MAX;
        }
    }
}

struct QDir {
    const float *q;      // sorted queries   (B, nqpad, 3)
    const int *qidx;     // their original indices
    const float *c;      // sorted candidates (B, ncpad, 3)
    const int *cidx;
    float *dist;         // (B, nq) in ORIGINAL query order
    int *idx;
    int nq, nqpad, nc, ncpad;
};
struct QArgs { QDir d[2]; };

__device__ __forceinline__ float dist3(float cx, float cy, float cz, float qx, float qy, float qz) {
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}

// one chunk of 8 wave-uniform candidates at sorted position k
__device__ __forceinline__ void eval_chunk(const float *__restrict__ c, const int *__restrict__ cidx, int k, float qx,
                                           float qy, float qz, float &best, int &bidx) {
    const float *__restrict__ ck = c + (size_t)k * 3;                 // wave-uniform -> scalar loads
    float d[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) d[u] = dist3(ck[u * 3 + 0], ck[u * 3 + 1], ck[u * 3 + 2], qx, qy, qz);
    float dm = d[0];
#pragma unroll
    for (int u = 1; u < CH; ++u) dm = fminf(dm, d[u]);
    if (__builtin_amdgcn_ballot_w64(dm <= best) != 0) {               // some lane improves or ties: resolve exactly
        const int *__restrict__ ik = cidx + k;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int oi = ik[u];
            const bool better = d[u] < best || (d[u] == best && oi < bidx);
            best = better ? d[u] : best;
            bidx = better ? oi : bidx;
        }
    }
}

__global__ __launch_bounds__(256) void nnq_kernel(QArgs args) {
    const QDir A = args.d[blockIdx.z];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = (blockIdx.x * 4 + wave) * 64;
    if (base >= A.nq) return;
    const float *__restrict__ q = A.q + (size_t)bi * A.nqpad * 3;
    const float *__restrict__ c = A.c + (size_t)bi * A.ncpad * 3;
    const int *__restrict__ cidx = A.cidx + (size_t)bi * A.ncpad;
    const int nc = A.nc;
    const int j = base + lane;
    const int jc = min(j, A.nq - 1);
    const float qx = q[jc * 3 + 0], qy = q[jc * 3 + 1], qz = q[jc * 3 + 2];

    // sorted position of the wave's median query among the candidates (two ballot rounds)
    const int last = min(base + 63, A.nq - 1);
    const float xmid = q[((base + last) >> 1) * 3];
    const int stride = (nc + 63) / 64;
    int pos;
    {
        const float x1 = c[(size_t)min(lane * stride, nc - 1) * 3];
        const int p1 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(x1 < xmid && lane * stride < nc));   // coarse cell
        const int lo = max(p1 - 1, 0) * stride;
        int cnt = 0;
        for (int o = 0; o < stride; o += 64) {
            const int kk = lo + o + lane;
            const float x2 = c[(size_t)min(kk, nc - 1) * 3];
            cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kk < nc && o + lane < stride && x2 < xmid));
        }
        pos = lo + cnt;
    }
    const int c0 = min(pos, nc - 1) & ~(CH - 1);

    float best = __builtin_inff();
    int bidx = INT_MAX;
    // ---- expand right
    for (int k = c0; k < A.ncpad; k += CH) {
        eval_chunk(c, cidx, k, qx, qy, qz, best, bidx);
        if (k + CH >= A.ncpad) break;
        const float xn = c[(size_t)(k + CH) * 3];                    // first x of the next chunk (uniform)
        const float dx = xn - qx;
        const bool done = xn > qx && dx * dx > best;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    }
    // ---- expand left
    for (int k = c0 - CH; k >= 0; k -= CH) {
        eval_chunk(c, cidx, k, qx, qy, qz, best, bidx);
        if (k == 0) break;
        const float xp = c[(size_t)(k - 1) * 3];                     // last x of the next chunk to the left
        const float dx = xp - qx;
        const bool done = xp < qx && dx * dx > best;
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    }
    if (j < A.nq) {
        const int o = A.qidx[(size_t)bi * A.nqpad + j];
        A.dist[(size_t)bi * A.nq + o] = best;
        A.idx[(size_t)bi * A.nq + o] = bidx;
    }
}

}  // namespace

extern "C" size_t dpf_nndistance_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return (size_t)b * ((size_t)pad8(n) + pad8(m)) * 16;
}

extern "C" int dpf_nndistance_ws(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                                 float *result2, int *result2_i, void *workspace, size_t workspace_bytes,
                                 dpf_stream_t stream) {
    if (b < 0 || n <= 0 || m <= 0) return DPF_EINVAL;
    if (b == 0) return 0;
    if (!xyz || !xyz2 || !result || !result_i || !result2 || !result2_i) return DPF_EINVAL;
    // outside the sort's range (or too small to be worth it): the brute-force kernel
    if (n > MAXN || m > MAXN || n < 64 || m < 64 || b > 65535 || !workspace ||
        workspace_bytes < dpf_nndistance_workspace_bytes(b, n, m))
        return dpf_nndistance(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, stream);
    hipStream_t s = (hipStream_t)stream;
    const int np = pad8(n), mp = pad8(m);
    float *sp1 = (float *)workspace;
    float *sp2 = sp1 + (size_t)b * np * 3;
    int *si1 = (int *)(sp2 + (size_t)b * mp * 3);
    int *si2 = si1 + (size_t)b * np;
    SortArgs sa;
    sa.s[0] = SortSet{xyz, sp1, si1, n, np};
    sa.s[1] = SortSet{xyz2, sp2, si2, m, mp};
    int np2 = 1;
    while (np2 < (n > m ? n : m)) np2 <<= 1;
    const int lds = np2 * 8;
    static int attr_lds = 0;
    if (lds > 65536 && lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void *)sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(sort_kernel, dim3(b, 2), dim3(1024), lds, s, sa);
    QArgs qa;
    qa.d[0] = QDir{sp1, si1, sp2, si2, result, result_i, n, np, m, mp};      // nndistance.cu:126
    qa.d[1] = QDir{sp2, si2, sp1, si1, result2, result2_i, m, mp, n, np};    // nndistance.cu:127
    const int nmax = n > m ? n : m;
    hipLaunchKernelGGL(nnq_kernel, dim3((nmax + 255) / 256, b, 2), dim3(256), 0, s, qa);
    return (int)hipGetLastError();
}
