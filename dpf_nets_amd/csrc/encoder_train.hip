// Training-mode PointNet cloud encoder for gfx950 (MI355X): forward with batch-statistics BatchNorm, the max over
// the points, and the full backward pass to the parameter gradients.
//
// Replaces, for model.train(),
//   PointNetCloudEncoder.forward           lib/networks/encoders.py:27-28
//     features = [SharedDot(no bias) . BatchNorm1d . ReLU] x 4,  3 -> 64 -> 128 -> 256 -> 512   (encoders.py:15-25)
//   torch.max(features, dim=2)[0]          lib/networks/models.py:131
// and what autograd derives from them (lib/networks/training.py:55).
//
// In training mode every BatchNorm normalises with statistics over all B*N points, so each layer ends in a grid-wide
// reduction and -- unlike the eval-mode kernel (encoder.hip) -- the stack cannot be one launch.  Each reduction is a
// kernel boundary; between boundaries the pre-BatchNorm outputs y_l (fp32, (B, C_l, Np), Np = N rounded up to 32)
// live in HBM: at cfg-2 that is 252 MB, a few percent of the part.  Every kernel here moves 3-4 bytes per bf16x3
// product triple and is HBM-bound; the design removes passes over the activations rather than flops:
//
//  forward   et_wpack     W_l as bf16 split MFMA B-operand fragments, both orientations (forward and W^T), one launch
//            et_l0        y0 = W0 x (fp32 FMA) and its statistics
//            et_bn_finish per-workgroup (sum, M2, count) -> mean, 1/std, folded scale/shift; running statistics
//            et_pgemm<FWD> y_l = W_l relu(BN(y_{l-1})): the BatchNorm + ReLU + bf16 split is the PROLOGUE on the operand
//                         registers (the activations a_l are never stored); the epilogue stores y_l and emits the
//                         per-workgroup statistics                                                  (l = 1, 2, 3)
//            et_pool      pooled[b, f] = relu(max_p BN(y3)), the argmax and y3 there -- from the per-tile extremes the layer-3
//                         epilogue leaves (no pass over y3)
//  backward  et_pool_bwd  the pooled gradient is a sparse d z3 (one point per (b, f)); its BatchNorm sums
//            et_bn_bwd_finish  d gamma, d beta and the coefficients of  d y_l = c0 dz_l + c1 + c2 y_l
//            et_pgemm<BWD> G = W_l^T d y_l (prologue forms d y_l from dz_l and y_l, for l = 3 from y_3 and the argmax);
//                         the epilogue masks with the ReLU of layer l-1, stores dz_{l-1}, emits its BatchNorm sums, and
//                         writes a_{l-1} = relu(BN(y_{l-1})) as bf16 hi/lo fragments with K = points -- the layout a lane
//                         of this epilogue already holds -- for the weight gradient
//            et_wgrad     dW_l = d y_l a_{l-1}^T, split over the points: d y_l is formed from the fp32 rows on the way
//                         from the global loads to LDS; et_wreduce adds the splits in a fixed order
//            et_wgrad0    dW_0 = d y_0 x^T (fp32)
//
// The forward contractions -- which decide every ReLU mask and the argmax -- run at the precision the caller asks for:
// bf16x6 (hi/mid/lo split, six products, fp32-class; the host side's default: a ReLU whose pre-activation is within
// the forward error of zero takes the other subgradient than the reference's, and at 1e-5 that happens to ~1e-5 of
// all ReLUs) or bf16x3.  The gradient contractions use hi/lo splits (hi*hi + hi*lo + lo*hi), fp32 accumulation.
// Every sum is taken in a fixed order (no floating-point atomics): the result is deterministic.
// d(input) is produced on request (et_dx; the clouds are normally data and it is skipped).
#include "flow_common.h"
#include "encoder_layout.h"
#include "graph_cache.h"

// Timing experiments only (results are garbage): -DET_ABLATE=<mask> removes parts of et_pgemm_kernel --
// 1 operand-row loads, 2 epilogue loads / stores, 4 MFMAs, 8 weight staging, 16 the prologue arithmetic.
#ifndef ET_ABLATE
#define ET_ABLATE 0
#endif

namespace {

constexpr int TCSUM = EC1 + EC2 + EC3 + EC4;                               // 960 features over the four layers
__host__ __device__ constexpr int t_coff(int l) { return l == 0 ? 0 : l == 1 ? EC1 : l == 2 ? EC1 + EC2 : EC1 + EC2 + EC3; }
constexpr int PW = 8;                       // waves (= 32-point tiles) per workgroup of the per-point kernels

struct Geo {
    int B, N, Np, tpc, ptiles, nwg;         // Np = padded points per cloud, tpc = tiles per cloud, nwg = ceil(ptiles / PW)
    long P;                                 // B * Np
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float half_add(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    return u2f(r[0]) + u2f(r[1]);
}
// 8 fp32 values -> NS bf16 fragments: truncated leading parts, the last one rounded (x = sum of the parts to 8 NS bits)
template <int NS>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&part)[NS]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float a = v[2 * d], b = v[2 * d + 1];
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) {
            float ra, rb;
            split_hi(a, ra); split_hi(b, rb);
            part[q][d] = pack_bf16_trunc(a, b);
            a = ra; b = rb;
        }
        part[NS - 1][d] = pack_bf16_rne(a, b);
    }
}

// ---- weights as B-operand fragments ---------------------------------------------------------------------------------
// out: [chunk][kstep KS][nt NT][part NS][lane 64][8 bf16]; element = Bm[k][n], k = 16 ks + 8 (lane >> 5) + j,
// n = 32 (chunk NT + nt) + (lane & 31), Bm[k][n] = W[k * sk + n * sn].  One launch packs all six matrices (blockIdx.y).
struct WPackJob { const float *W; uint8_t *out; int sk, sn, KS, NT, nchunk, NS; };
struct WPackArgs { WPackJob job[6]; };
template <int NS>
__device__ __forceinline__ void wpack_job(const WPackJob &q) {
    const int total = q.nchunk * q.KS * q.NT * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, frag = idx >> 6;
        const int nt = frag % q.NT, ks = (frag / q.NT) % q.KS, chunk = frag / (q.NT * q.KS);
        const int n = 32 * (chunk * q.NT + nt) + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = q.W[(size_t)(k0 + j) * q.sk + (size_t)n * q.sn];
        u32x4 part[NS];
        split8<NS>(v, part);
        uint8_t *o = q.out + ((size_t)frag * NS) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < NS; ++t) *(u32x4 *)(o + t * 1024) = part[t];
    }
}
__global__ __launch_bounds__(256) void et_wpack_kernel(WPackArgs a) {
    const WPackJob &q = a.job[blockIdx.y];
    if (q.NS == 3) wpack_job<3>(q); else wpack_job<2>(q);
}

// ---- layer 0: y0 = W0 x --------------------------------------------------------------------------------------------
// Statistics travel as (sum, M2 about the group's own mean, count): tile -> workgroup here, workgroup -> batch in
// et_bn_finish (Chan's pairwise combination), so a variance far below the squared mean keeps its digits.
__device__ __forceinline__ int tile_count(const Geo &g, int ptile) {     // valid points of a 32-point tile
    if (ptile >= g.ptiles) return 0;
    const int left = g.N - (ptile % g.tpc) * TILE;
    return left >= TILE ? TILE : (left > 0 ? left : 0);
}
__global__ __launch_bounds__(256) void et_l0_kernel(Geo g, const float *__restrict__ W0, const float *__restrict__ x,
                                                     float *__restrict__ y0, float *__restrict__ part, float *__restrict__ cnt) {
    __shared__ float w[EC1 * EC0];
    __shared__ float tile[32][257];
    __shared__ float ts[PW][2][EC1];
    const int tid = threadIdx.x;
    if (tid < EC1 * EC0) w[tid] = W0[tid];
    const int ptile = blockIdx.x * PW + (tid >> 5), pl = tid & 31;
    const bool tile_ok = ptile < g.ptiles;
    const int b = tile_ok ? ptile / g.tpc : 0, p = tile_ok ? (ptile % g.tpc) * TILE + pl : 0;
    const bool live = tile_ok && p < g.N;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (live) {
        const float *xc = x + (size_t)b * 3 * g.N + p;
        x0 = xc[0]; x1 = xc[g.N]; x2 = xc[2 * (size_t)g.N];
    }
    float *yo = y0 + (size_t)b * EC1 * g.Np + p;
    const int rf = tid & 31, rt = tid >> 5, rn = tile_count(g, blockIdx.x * PW + rt);      // reduction role: feature, tile
    __syncthreads();
    for (int half = 0; half < 2; ++half) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int f = 32 * half + i;
            const float v = live ? fmaf(w[3 * f + 2], x2, fmaf(w[3 * f + 1], x1, w[3 * f] * x0)) : 0.f;
            if (tile_ok) yo[(size_t)f * g.Np] = v;
            tile[i][tid] = v;
        }
        __syncthreads();
        float s = 0.f, m2 = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += tile[rf][32 * rt + q];
        const float m = rn > 0 ? s / (float)rn : 0.f;
        for (int q = 0; q < rn; ++q) { const float d = tile[rf][32 * rt + q] - m; m2 = fmaf(d, d, m2); }
        ts[rt][0][32 * half + rf] = s;
        ts[rt][1][32 * half + rf] = m2;
        __syncthreads();
    }
    if (tid < EC1) {
        float n = 0.f, s = 0.f;
        for (int t = 0; t < PW; ++t) { n += (float)tile_count(g, blockIdx.x * PW + t); s += ts[t][0][tid]; }
        const float mean = n > 0.f ? s / n : 0.f;
        float m2 = 0.f;
        for (int t = 0; t < PW; ++t) {
            const float nt = (float)tile_count(g, blockIdx.x * PW + t);
            if (nt > 0.f) { const float d = ts[t][0][tid] / nt - mean; m2 += ts[t][1][tid] + nt * d * d; }
        }
        part[((size_t)blockIdx.x * 2) * EC1 + tid] = s;
        part[((size_t)blockIdx.x * 2 + 1) * EC1 + tid] = m2;
        if (tid == 0) cnt[blockIdx.x] = n;
    }
}

// ---- BatchNorm statistics: partials -> folded parameters -----------------------------------------------------------
// part: [nwg][2][C] = per-workgroup (sum, M2 about the workgroup mean); cnt[nwg] = its number of points.
// bnp: [4][C] = scale (gamma / std), shift (beta - mean * scale), mean, 1 / std.  One workgroup per 64 features, 16
// slices of the partials per feature, combined in a fixed order.
constexpr int FS = 64, FF = 16;            // slices of the partials per feature, features per workgroup
__global__ __launch_bounds__(FF * FS) void et_bn_finish_kernel(int nwg, int C, double count, const float *__restrict__ part,
                                                                const float *__restrict__ cnt, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float *__restrict__ bnp,
                                                                float *__restrict__ run_mean, float *__restrict__ run_var, float momentum,
                                                                float *__restrict__ bstat) {
    __shared__ double rn[FS][FF], rs[FS][FF], rm[FS][FF];
    const int f = threadIdx.x % FF, sl = threadIdx.x / FF, c = blockIdx.x * FF + f;
    double N = 0.0, S = 0.0, M2 = 0.0;                       // this thread's groups, M2 about their own mean S / N
    for (int w0 = sl; w0 < nwg; w0 += 4 * FS) {              // (one trip up to 256 workgroups) all loads in flight at once
        float vn[4], v1[4], v2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int w = w0 + q * FS;
            const bool ok = w < nwg;
            vn[q] = ok ? cnt[w] : 0.f;
            v1[q] = ok ? part[((size_t)w * 2) * C + c] : 0.f;
            v2[q] = ok ? part[((size_t)w * 2 + 1) * C + c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double n = (double)vn[q];
            if (n > 0.0) {                                   // Chan's pairwise update
                const double d = (double)v1[q] / n - (N > 0.0 ? S / N : 0.0);
                M2 += (double)v2[q] + d * d * N * n / (N + n);
                N += n;
                S += (double)v1[q];
            }
        }
    }
    rn[sl][f] = N; rs[sl][f] = S; rm[sl][f] = M2;
    __syncthreads();
    for (int step = FS / 2; step >= 1; step >>= 1) {         // pairwise (Chan) tree over the slices, fixed order
        if (sl < step) {
            const double na = rn[sl][f], nb = rn[sl + step][f];
            if (nb > 0.0) {
                if (na > 0.0) {
                    const double d = rs[sl + step][f] / nb - rs[sl][f] / na;
                    rm[sl][f] += rm[sl + step][f] + d * d * na * nb / (na + nb);
                } else {
                    rm[sl][f] = rm[sl + step][f];
                }
                rn[sl][f] = na + nb;
                rs[sl][f] += rs[sl + step][f];
            }
        }
        __syncthreads();
    }
    if (sl != 0) return;
    const double mean = rs[0][f] / count, t = rm[0][f];
    const double var = t / count;
    const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const float sc = gamma[c] * rstd;
    bnp[c] = sc;
    bnp[C + c] = beta[c] - (float)mean * sc;
    bnp[2 * C + c] = (float)mean;
    bnp[3 * C + c] = rstd;
    if (bstat != nullptr) { bstat[c] = (float)mean; bstat[C + c] = (float)var; }
    if (run_mean != nullptr) {      // torch.nn.BatchNorm1d: running_var takes the unbiased estimate
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// backward: partials of (sum dz, sum dz * yhat) -> d beta, d gamma, coef [3][C]:  d y = c0 dz + c1 + c2 y
__global__ __launch_bounds__(FF * FS) void et_bn_bwd_finish_kernel(int nwg, int C, double count, const float *__restrict__ part,
                                                                    const float *__restrict__ bnp, float *__restrict__ coef,
                                                                    float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double red[2][FS][FF];
    const int f = threadIdx.x % FF, sl = threadIdx.x / FF, c = blockIdx.x * FF + f;
    double t1 = 0.0, t2 = 0.0;
    for (int w0 = sl; w0 < nwg; w0 += 4 * FS) {
        float v1[4], v2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int w = w0 + q * FS;
            v1[q] = w < nwg ? part[((size_t)w * 2) * C + c] : 0.f;
            v2[q] = w < nwg ? part[((size_t)w * 2 + 1) * C + c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { t1 += (double)v1[q]; t2 += (double)v2[q]; }
    }
    red[0][sl][f] = t1;
    red[1][sl][f] = t2;
    __syncthreads();
    if (sl != 0) return;
    t1 = 0.0; t2 = 0.0;
    for (int q = 0; q < FS; ++q) { t1 += red[0][q][f]; t2 += red[1][q][f]; }
    dbeta[c] = (float)t1;
    dgamma[c] = (float)t2;
    const float sc = bnp[c], mean = bnp[2 * C + c], rstd = bnp[3 * C + c];
    const float m1 = (float)(t1 / count), m2 = (float)(t2 / count);
    const float c2 = -sc * m2 * rstd;
    coef[c] = sc;
    coef[C + c] = -sc * m1 - c2 * mean;
    coef[2 * C + c] = c2;
}

// ---- per-point GEMM: out[point][n] = sum_k op(in)[point][k] Bm[k][n] -------------------------------------------------
// A workgroup = 8 waves = 8 tiles of 32 points; it produces NT * 32 output features (chunk blockIdx -> `chunk`) of them.
// Four waves per SIMD (<= 128 VGPRs, two workgroups per CU): these kernels move 3-4 bytes per bf16x3 flop-triple and are
// HBM-bound, so what matters is bytes in flight -- each wave keeps the operand rows of the next two k-steps in
// registers (plain loads, counted in order by vmcnt, never drained by a barrier), and the weight fragments of the next
// chunk travel global -> registers -> LDS one chunk ahead (one workgroup barrier per chunk of two k-steps).
// Workgroup ids are remapped so that the `nch` chunks of the same eight tiles run on the same XCD (ids equal mod 8)
// back to back: their operand rows are read from HBM once and from that XCD's L2 afterwards.
enum { FWD = 0, BWD = 1, BWD_SPARSE = 2 };
struct PArgs {
    Geo g;
    const float *yin;       // (B, K, Np): FWD y_{l-1}; BWD y_l
    const float *dzin;      // (B, K, Np): BWD dz_l
    const float *pin;       // FWD: bnp of layer l-1 [4][K]; BWD: coef of layer l [3][K]
    const int *arg;         // BWD_SPARSE: (B, K) argmax point
    const float *gz;        // BWD_SPARSE: (B, K) masked pooled gradient
    const uint8_t *wpk;     // [nch][KS][NT][NS][64][16 B]
    float *out;             // (B, Ntot, Np): FWD y_l; BWD dz_{l-1}
    const float *yprev;     // BWD: (B, Ntot, Np) y_{l-1}
    const float *bnprev;    // BWD: bnp of layer l-1 [4][Ntot]
    float *part;            // [nwg][2][Ntot]
    uint8_t *apk;           // BWD: a_{l-1} = relu(BN(y_{l-1})) as K = points fragments (operand of dW_l), layout at WArgs
    float *tmax, *tmin;     // FWD, last layer: [ptile][Ntot] largest / smallest y of every tile's valid points (for et_pool)
    int Ntot, nch;
};

template <int KS, int NT, int MODE, int NS, bool POOL = false>
__global__ __launch_bounds__(PW * 64, 4) void et_pgemm_kernel(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef Terms<NS> TT;
    constexpr int K = KS * 16, KC = 2, NCH = KS / KC, BUF = KC * NT * NS * 1024, NW = BUF / (PW * 1024);
    constexpr int NG = NT < 2 ? NT : 2;                      // output tiles whose MFMAs are interleaved
    static_assert(KS % KC == 0 && BUF % (PW * 1024) == 0 && NW >= 1, "chunking");
    constexpr int NPAR = MODE == FWD ? 2 : 3;
    uint8_t *l_w = smem;                                     // [2][BUF]
    float *l_par = (float *)(smem + 2 * BUF);                // [NPAR][K]
    float *l_red = l_par + NPAR * K;                         // [PW][2][NT * 32]
    int *l_arg = (int *)(l_red + PW * 2 * NT * 32);          // BWD_SPARSE: [PW][K] argmax point of the wave's cloud
    float *l_gz = (float *)(l_arg + PW * K);                 //             [PW][K] its masked pooled gradient

    const Geo &g = a.g;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = blockIdx.x;
    const int tg = (id / (8 * a.nch)) * 8 + (id & 7), chunk = (id >> 3) % a.nch;
    if (tg >= g.nwg) return;
    int ptile = tg * PW + wave;
    const bool tile_ok = ptile < g.ptiles;
    if (!tile_ok) ptile = g.ptiles - 1;
    const int b = ptile / g.tpc, p0 = (ptile % g.tpc) * TILE;
    const bool live = tile_ok && p0 + pl < g.N;              // this lane's point exists

    const uint8_t *wsrc = a.wpk + (size_t)chunk * KS * NT * NS * 1024 + (size_t)wave * NW * 1024 + lane * 16;
    u32x4 wreg[NW];
    auto wload = [&](int c) {                                // this wave's share of chunk c of the weight stream
#pragma unroll
        for (int i = 0; i < NW; ++i) wreg[i] = *(const u32x4 *)(wsrc + (size_t)c * BUF + i * 1024);
    };
    auto wstore = [&](int c) {
#pragma unroll
        for (int i = 0; i < NW; ++i) *(u32x4 *)(l_w + (c & 1) * BUF + (wave * NW + i) * 1024 + lane * 16) = wreg[i];
    };
    wload(0);
    for (int i = threadIdx.x; i < NPAR * K; i += PW * 64) l_par[i] = a.pin[i];
    if (MODE == BWD_SPARSE) {
#pragma unroll
        for (int i = 0; i < K / 256; ++i) {
            *(int4 *)(l_arg + wave * K + i * 256 + lane * 4) = *(const int4 *)(a.arg + (size_t)b * K + i * 256 + lane * 4);
            *(f32x4 *)(l_gz + wave * K + i * 256 + lane * 4) = *(const f32x4 *)(a.gz + (size_t)b * K + i * 256 + lane * 4);
        }
    }
    wstore(0);
    if (NCH > 1) wload(1);

    // operand rows of this lane: features 16 s + 8 h + j of point p0 + pl
    const size_t in_base = (size_t)b * K * g.Np + p0 + pl;
    float ry[2][8], rz[2][8];
    auto fetch = [&](int s, float (&y)[8], float (&z)[8]) {
        const size_t o = in_base + (size_t)(16 * s + 8 * h) * g.Np;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (ET_ABLATE & 1) { y[j] = 1.f + (float)o; z[j] = 2.f; continue; }
            y[j] = a.yin[o + (size_t)j * g.Np];
            if (MODE == BWD) z[j] = a.dzin[o + (size_t)j * g.Np];
        }
    };
    auto operand = [&](int s, const float (&y)[8], const float (&z)[8], u32x4 (&frag)[NS]) {   // prologue -> A fragment parts
        const int f0 = 16 * s + 8 * h;
        float v[8];
        if (ET_ABLATE & 16) {
#pragma unroll
            for (int q = 0; q < NS; ++q) frag[q] = u32x4{f2u(y[0]), f2u(y[1]), f2u(y[2]) + q, f2u(y[3]) ^ f2u(y[4]) ^ f2u(y[5]) ^ f2u(y[6]) ^ f2u(y[7]) ^ f2u(z[0])};
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 c0 = *(const f32x4 *)(l_par + f0 + 4 * q), c1 = *(const f32x4 *)(l_par + K + f0 + 4 * q);
            f32x4 c2 = c1, zz = c1;
            if (MODE != FWD) c2 = *(const f32x4 *)(l_par + 2 * K + f0 + 4 * q);
            if (MODE == BWD) zz = f32x4{z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
            if (MODE == BWD_SPARSE) {
                const int4 ap = *(const int4 *)(l_arg + wave * K + f0 + 4 * q);
                const f32x4 gv = *(const f32x4 *)(l_gz + wave * K + f0 + 4 * q);
                const int p = p0 + pl;
                zz = f32x4{ap.x == p ? gv.x : 0.f, ap.y == p ? gv.y : 0.f, ap.z == p ? gv.z : 0.f, ap.w == p ? gv.w : 0.f};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                float r;
                if (MODE == FWD) r = fmaxf(fmaf(y[j], c0[e], c1[e]), 0.f);
                else r = fmaf(c0[e], zz[e], fmaf(c2[e], y[j], c1[e]));
                v[j] = live ? r : 0.f;
            }
        }
        split8<NS>(v, frag);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    fetch(0, ry[0], rz[0]);
    fetch(1, ry[1], rz[1]);
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();                                     // chunk c is in buffer c & 1; nobody reads the other buffer any more
        if (c + 1 < NCH && !(ET_ABLATE & 8)) {
            wstore(c + 1);
            if (c + 2 < NCH) wload(c + 2);
        }
        const uint8_t *wb = l_w + (c & 1) * BUF;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int s = c * KC + kc;
            u32x4 af[NS];
            operand(s, ry[kc], rz[kc], af);
            if (s + 2 < KS) fetch(s + 2, ry[kc], rz[kc]);
#pragma unroll
            for (int ng = 0; ng < NT; ng += NG) {
                u32x4 wf[NG][NS];
#pragma unroll
                for (int i = 0; i < NG; ++i)
#pragma unroll
                    for (int q = 0; q < NS; ++q)
                        wf[i][q] = *(const u32x4 *)(wb + ((kc * NT + ng + i) * NS + q) * 1024 + lane * 16);
#pragma unroll
                for (int term = 0; term < TT::N; ++term)
#pragma unroll
                    for (int i = 0; i < NG; ++i) {
                        if (ET_ABLATE & 4) { acc[ng + i][term] += u2f(af[TT::A[term]][0] ^ wf[i][TT::B[term]][1]); continue; }
                        acc[ng + i] = mfma(af[TT::A[term]], wf[i][TT::B[term]], acc[ng + i]);
                    }
            }
        }
    }
    if (ET_ABLATE & 2) {
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[nt][r];
        if (t == 12345.f) a.part[0] = t;
        return;
    }

    // epilogue: accumulator register r = point p0 + (r & 3) + 8 (r >> 2) + 4 h, lane column = feature n0 + 32 nt + pl
    const int n0 = chunk * NT * 32;
    const int nw = tile_ok ? tile_count(g, ptile) : 0;       // valid points of this wave's tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int f = n0 + 32 * nt + pl;
        const size_t o = ((size_t)b * a.Ntot + f) * g.Np + p0 + 4 * h;
        float s1 = 0.f, s2 = 0.f;
        if (MODE == FWD) {          // (sum, M2 about the tile's own mean); padded points hold exact zeros
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[nt][r];
            s1 = half_add(s1);
            const float m = nw > 0 ? s1 / (float)nw : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[nt][r] - m;
                const bool valid = nw == TILE || (r & 3) + 8 * (r >> 2) + 4 * h < nw;
                s2 = valid ? fmaf(d, d, s2) : s2;
            }
            s2 = half_add(s2);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (tile_ok) *(f32x4 *)(a.out + o + 8 * q) = f32x4{acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
            if (POOL) {             // extremes of the tile's valid points: et_pool picks the winning tile per (cloud, feature) from them
                float mx = -__builtin_inff(), mn = __builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool valid = nw == TILE || (r & 3) + 8 * (r >> 2) + 4 * h < nw;
                    mx = valid ? fmaxf(mx, acc[nt][r]) : mx;
                    mn = valid ? fminf(mn, acc[nt][r]) : mn;
                }
                const auto rx = __builtin_amdgcn_permlane32_swap(f2u(mx), f2u(mx), false, false);
                const auto rn = __builtin_amdgcn_permlane32_swap(f2u(mn), f2u(mn), false, false);
                if (tile_ok && !h) {
                    a.tmax[(size_t)ptile * a.Ntot + f] = fmaxf(u2f(rx[0]), u2f(rx[1]));
                    a.tmin[(size_t)ptile * a.Ntot + f] = fminf(u2f(rn[0]), u2f(rn[1]));
                }
            }
        } else {
            const float sc = a.bnprev[f], sh = a.bnprev[a.Ntot + f], mean = a.bnprev[2 * a.Ntot + f], rstd = a.bnprev[3 * a.Ntot + f];
            f32x4 y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = *(const f32x4 *)(a.yprev + o + 8 * q);      // all four in flight before the first store
            const long ks0 = ((long)b * g.Np + p0) / 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = fmaf(y[q][e], sc, sh);
                    v[e] = z > 0.f ? v[e] : 0.f;
                    s1 += v[e];
                    s2 = fmaf(v[e], (y[q][e] - mean) * rstd, s2);
                    y[q][e] = (nw == TILE || 4 * h + 8 * q + e < nw) ? fmaxf(z, 0.f) : 0.f;      // a_{l-1}; padded points give 0
                }
                if (tile_ok) *(f32x4 *)(a.out + o + 8 * q) = v;
            }
            if (tile_ok) {          // k-step j2 of the tile: this lane (row pl, k group h) holds points 16 j2 + 8 (i >> 2) + 4 h + (i & 3)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const float v8[8] = {y[2 * j2][0], y[2 * j2][1], y[2 * j2][2], y[2 * j2][3], y[2 * j2 + 1][0], y[2 * j2 + 1][1], y[2 * j2 + 1][2], y[2 * j2 + 1][3]};
                    u32x4 hl[2];
                    split8<2>(v8, hl);
                    uint8_t *dst = a.apk + ((((size_t)(n0 / 32 + nt)) * (g.P / 16) + ks0 + j2) * 2) * 1024 + lane * 16;
                    *(u32x4 *)dst = hl[0];
                    *(u32x4 *)(dst + 1024) = hl[1];
                }
            }
            s1 = half_add(s1); s2 = half_add(s2);
        }
        if (!tile_ok) { s1 = 0.f; s2 = 0.f; }
        if (!h) {
            l_red[(wave * 2 + 0) * (NT * 32) + 32 * nt + pl] = s1;
            l_red[(wave * 2 + 1) * (NT * 32) + 32 * nt + pl] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x < NT * 32) {
        const int f = threadIdx.x;
        float s = 0.f, t = 0.f;
        if (MODE == FWD) {
            float n = 0.f;
#pragma unroll
            for (int w = 0; w < PW; ++w) { n += (float)tile_count(g, tg * PW + w); s += l_red[(w * 2) * (NT * 32) + f]; }
            const float mean = n > 0.f ? s / n : 0.f;
#pragma unroll
            for (int w = 0; w < PW; ++w) {
                const float nt_ = (float)tile_count(g, tg * PW + w);
                if (nt_ > 0.f) { const float d = l_red[(w * 2) * (NT * 32) + f] / nt_ - mean; t += l_red[(w * 2 + 1) * (NT * 32) + f] + nt_ * d * d; }
            }
        } else {
#pragma unroll
            for (int w = 0; w < PW; ++w) { s += l_red[(w * 2) * (NT * 32) + f]; t += l_red[(w * 2 + 1) * (NT * 32) + f]; }
        }
        a.part[((size_t)tg * 2) * a.Ntot + n0 + f] = s;
        a.part[((size_t)tg * 2 + 1) * a.Ntot + n0 + f] = t;
    }
}

// ---- max over the points -------------------------------------------------------------------------------------------
// pooled[b, f] = relu(max_p BN(y3)[b, f, p]).  BN is monotone in y (increasing for scale >= 0, decreasing otherwise), so
// the winner is the point with the largest (smallest) y: the layer-3 GEMM's epilogue left every tile's extremes, a thread
// per (cloud, feature) finds the first winning tile and then the first winning point inside it -- 64 + 32 values read
// instead of the cloud's whole row of y3.
__global__ __launch_bounds__(64) void et_pool_kernel(Geo g, const float *__restrict__ y3, const float *__restrict__ tmax,
                                                      const float *__restrict__ tmin, const float *__restrict__ bnp,
                                                      float *__restrict__ pooled, int *__restrict__ arg, float *__restrict__ yarg) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= g.B * EC4) return;
    const int b = row / EC4, f = row % EC4;
    const float sc = bnp[f], sh = bnp[EC4 + f];
    const bool up = sc >= 0.f;
    const float *ext = (up ? tmax : tmin) + (size_t)b * g.tpc * EC4 + f;
    float best = up ? -__builtin_inff() : __builtin_inff();
    int bt = 0;
    for (int t0 = 0; t0 < g.tpc; t0 += 16) {                 // sixteen independent loads in flight, then the compares in tile order
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = t0 + q < g.tpc ? ext[(size_t)(t0 + q) * EC4] : best;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (up ? v[q] > best : v[q] < best) { best = v[q]; bt = t0 + q; }
    }
    const float *yr = y3 + (size_t)row * g.Np + bt * TILE;
    f32x4 r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = *(const f32x4 *)(yr + 4 * q);
    int bi = -1;
#pragma unroll
    for (int q = 7; q >= 0; --q)
#pragma unroll
        for (int e = 3; e >= 0; --e)
            if (r[q][e] == best && bt * TILE + 4 * q + e < g.N) bi = bt * TILE + 4 * q + e;      // descending: the first one wins
    pooled[row] = fmaxf(fmaf(best, sc, sh), 0.f);
    arg[row] = bi;
    yarg[row] = best;
}

// the pooled gradient as the sparse dz3: gz[b, f] = g[b, f] [pooled > 0] at point arg[b, f]; its BatchNorm sums
__global__ __launch_bounds__(256) void et_pool_bwd_kernel(int B, const float *__restrict__ gp, const float *__restrict__ pooled,
                                                           const float *__restrict__ yarg, const float *__restrict__ bnp,
                                                           float *__restrict__ gz, float *__restrict__ part) {
    __shared__ float red[2][4][64];
    const int fl = threadIdx.x & 63, sl = threadIdx.x >> 6, f = blockIdx.x * 64 + fl;
    const float mean = bnp[2 * EC4 + f], rstd = bnp[3 * EC4 + f];
    float s1 = 0.f, s2 = 0.f;
    for (int b = sl; b < B; b += 4) {
        const float v = pooled[b * EC4 + f] > 0.f ? gp[b * EC4 + f] : 0.f;
        gz[b * EC4 + f] = v;
        s1 += v;
        s2 = fmaf(v, (yarg[b * EC4 + f] - mean) * rstd, s2);
    }
    red[0][sl][fl] = s1;
    red[1][sl][fl] = s2;
    __syncthreads();
    if (sl == 0) {
        part[f] = (red[0][0][fl] + red[0][1][fl]) + (red[0][2][fl] + red[0][3][fl]);
        part[EC4 + f] = (red[1][0][fl] + red[1][1][fl]) + (red[1][2][fl] + red[1][3][fl]);
    }
}

// ---- weight gradient: part[kchunk][m][n] = sum over the chunk's points of d y[m][p] a[n][p] -----------------------------
// The a operand arrives as K = points fragments (written by the per-point GEMM's epilogue).  The d y operand is formed
// HERE from the fp32 rows: d y = c0 dz + c1 + c2 y (MODE BWD), dz = gz at the argmax point (BWD_SPARSE), on the way from
// the global loads to LDS -- each element is transformed once per workgroup, ~100 VALU instructions per wave next to 48 MFMAs.
// K = points fragment stream: [feature tile C/32][k-step P/16][part hi, lo][lane 64][8 bf16]; lane (row, kg) holds, for
// feature 32 ft + row, the points 16 s + 8 (i >> 2) + 4 kg + (i & 3), i = 0..7 (any order of a k-step's 16 points works
// as long as both operands use it; this is the order a lane of the per-point GEMM's epilogue holds).  Padded points are 0.
struct WArgs {
    Geo g;
    const float *y, *dz, *coef;     // (B, M, Np) rows of layer l; coef [3][M]
    const int *arg;                 // BWD_SPARSE: (B, M)
    const float *gz;
    const uint8_t *Bm;              // packed K = points operand a_{l-1}: [tile][PS][2][64][16 B]
    float *part;                    // [nchunk][M][Ncols]
    long PS;
    int ks_chunk, M, Ncols;
};

template <int WM, int WN, int GM, int GN, int MODE>
__global__ __launch_bounds__(GM * GN * 64) void et_wgrad_kernel(WArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NWV = GM * GN, BM = GM * WM, BN = GN * WN, KC = 2;
    constexpr int STAGE = KC * (BM + BN) * 2048;
    constexpr int NA = KC * BM / NWV, NB = KC * BN * 2 / NWV;      // per wave and stage: d y fragment pairs to form, a fragments to copy
    static_assert(KC * BM % NWV == 0 && KC * BN * 2 % NWV == 0, "staging");
    const Geo &g = a.g;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / GN, wn = wave % GN;
    const long s_begin = (long)blockIdx.x * a.ks_chunk;
    const long s_end_l = s_begin + a.ks_chunk < a.PS ? s_begin + a.ks_chunk : a.PS;
    const int nst = (int)((s_end_l - s_begin + KC - 1) / KC);
    const int rb = blockIdx.y, cb = blockIdx.z;

    // a stage = KC k-steps of all the block's tiles in LDS: [kc][tile BM + BN][part]; it travels global -> registers ->
    // (transform) -> LDS one stage ahead of its use (plain loads: the vmcnt waits are exact and no barrier drains them)
    f32x4 ry[NA][2], rd[NA][2];
    int rarg[NA];
    float rgz[NA];
    int rp[NA];                                               // first point (within its cloud) of the lane's 8, or -1
    float cf[NA][3];
    u32x4 rb_[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int tile = (wave + i * NWV) % BM, f = (rb * BM + tile) * 32 + pl;
        cf[i][0] = a.coef[f]; cf[i][1] = a.coef[a.M + f]; cf[i][2] = a.coef[2 * a.M + f];
    }
    auto sload = [&](int st) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = wave + i * NWV, tile = k % BM, kc = k / BM;
            long s = s_begin + (long)st * KC + kc;
            if (s >= a.PS) s = a.PS - 1;                      // tail: a repeated k-step, not used
            const long q = 16 * s + 4 * h;
            const int b = (int)(q / g.Np), p = (int)(q % g.Np), f = (rb * BM + tile) * 32 + pl;
            const size_t o = ((size_t)b * a.M + f) * g.Np + p;
            ry[i][0] = *(const f32x4 *)(a.y + o); ry[i][1] = *(const f32x4 *)(a.y + o + 8);
            if (MODE == BWD) { rd[i][0] = *(const f32x4 *)(a.dz + o); rd[i][1] = *(const f32x4 *)(a.dz + o + 8); }
            if (MODE == BWD_SPARSE) { rarg[i] = a.arg[(size_t)b * a.M + f]; rgz[i] = a.gz[(size_t)b * a.M + f]; }
            rp[i] = p;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = wave + i * NWV, part = k & 1, tile = (k >> 1) % BN, kc = (k >> 1) / BN;
            long s = s_begin + (long)st * KC + kc;
            if (s >= a.PS) s = a.PS - 1;
            rb_[i] = *(const u32x4 *)(a.Bm + ((((size_t)(cb * BN + tile)) * a.PS + s) * 2 + part) * 1024 + lane * 16);
        }
    };
    auto sstore = [&](int st) {
        uint8_t *dst = smem + (st & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = wave + i * NWV, tile = k % BM, kc = k / BM;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float yv = ry[i][j >> 2][j & 3];
                const int pj = rp[i] + (j & 3) + 8 * (j >> 2);
                float dzv;
                if (MODE == BWD) dzv = rd[i][j >> 2][j & 3];
                else dzv = rarg[i] == pj ? rgz[i] : 0.f;
                const float r = fmaf(cf[i][0], dzv, fmaf(cf[i][2], yv, cf[i][1]));
                v[j] = pj < g.N ? r : 0.f;
            }
            u32x4 hl[2];
            split8<2>(v, hl);
            *(u32x4 *)(dst + ((kc * (BM + BN) + tile) * 2 + 0) * 1024 + lane * 16) = hl[0];
            *(u32x4 *)(dst + ((kc * (BM + BN) + tile) * 2 + 1) * 1024 + lane * 16) = hl[1];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = wave + i * NWV, part = k & 1, tile = (k >> 1) % BN, kc = (k >> 1) / BN;
            *(u32x4 *)(dst + ((kc * (BM + BN) + BM + tile) * 2 + part) * 1024 + lane * 16) = rb_[i];
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sload(0);
    sstore(0);
    if (nst > 1) sload(1);
    for (int st = 0; st < nst; ++st) {
        __syncthreads();
        if (st + 1 < nst) {
            sstore(st + 1);
            if (st + 2 < nst) sload(st + 2);
        }
        const uint8_t *sb = smem + (st & 1) * STAGE;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (s_begin + (long)st * KC + kc < s_end_l) {
                const uint8_t *kb = sb + kc * (BM + BN) * 2048;
                u32x4 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    ah[i] = *(const u32x4 *)(kb + ((wm * WM + i) * 2 + 0) * 1024 + lane * 16);
                    al[i] = *(const u32x4 *)(kb + ((wm * WM + i) * 2 + 1) * 1024 + lane * 16);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    bh[j] = *(const u32x4 *)(kb + ((BM + wn * WN + j) * 2 + 0) * 1024 + lane * 16);
                    bl[j] = *(const u32x4 *)(kb + ((BM + wn * WN + j) * 2 + 1) * 1024 + lane * 16);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = mfma(al[i], bh[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = mfma(ah[i], bl[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = mfma(ah[i], bh[j], acc[i][j]);
            }
        }
    }
    float *po = a.part + (size_t)blockIdx.x * a.M * a.Ncols;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = ((cb * BN + wn * WN + j) * 32) + pl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowi = (rb * BM + wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                po[(size_t)rowi * a.Ncols + col] = acc[i][j][r];
            }
        }
}

// out[e] = sum over the chunks of part[chunk][e]: four slices of the chunks per element, combined in a fixed order
__global__ __launch_bounds__(256) void et_wreduce_kernel(int nchunk, int E4, const f32x4 *__restrict__ part, f32x4 *__restrict__ out) {
    __shared__ f32x4 red[4][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6, e = blockIdx.x * 64 + el;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = sl; c < nchunk; c += 4) {
        const f32x4 v = part[(size_t)c * E4 + e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0) {
        const f32x4 a0 = red[0][el], a1 = red[1][el], a2 = red[2][el], a3 = red[3][el];
        out[e] = f32x4{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
    }
}

// dW0[f][k] = sum_p d y0[f][p] x[k][p], d y0 = c0 dz0 + c1 + c2 y0: one workgroup per (feature, cloud) -> part[b][f][3 (+1 pad)]
__global__ __launch_bounds__(256) void et_wgrad0_kernel(Geo g, const float *__restrict__ y0, const float *__restrict__ dz0,
                                                         const float *__restrict__ coef, const float *__restrict__ x,
                                                         float *__restrict__ part) {
    __shared__ float red[4][3];
    const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float c0 = coef[f], c1 = coef[EC1 + f], c2 = coef[2 * EC1 + f];
    const float *yr = y0 + ((size_t)b * EC1 + f) * g.Np, *dr = dz0 + ((size_t)b * EC1 + f) * g.Np;
    const float *xc = x + (size_t)b * 3 * g.N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int p = threadIdx.x; p < g.N; p += 256) {
        const float dy = fmaf(c0, dr[p], fmaf(c2, yr[p], c1));
        s0 = fmaf(dy, xc[p], s0);
        s1 = fmaf(dy, xc[g.N + p], s1);
        s2 = fmaf(dy, xc[2 * (size_t)g.N + p], s2);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((size_t)b * EC1 + f) * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(768) void et_w0reduce_kernel(int B, const float *__restrict__ part, float *__restrict__ out) {
    __shared__ float red[4][192];
    const int e = threadIdx.x % 192, sl = threadIdx.x / 192;
    float s = 0.f;
    for (int b = sl; b < B; b += 4) s += part[(size_t)b * EC1 * 3 + e];
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// d x[b][k][p] = sum_f W0[f][k] d y0[f][p] (only when the caller asks for the input gradient)
__global__ __launch_bounds__(256) void et_dx_kernel(Geo g, const float *__restrict__ y0, const float *__restrict__ dz0,
                                                     const float *__restrict__ coef, const float *__restrict__ W0, float *__restrict__ dx) {
    __shared__ float w[EC1 * EC0], c[3 * EC1];
    if (threadIdx.x < EC1 * EC0) w[threadIdx.x] = W0[threadIdx.x];
    if (threadIdx.x < 3 * EC1) c[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= g.N) return;
    const float *yr = y0 + (size_t)b * EC1 * g.Np + p, *dr = dz0 + (size_t)b * EC1 * g.Np + p;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 8
    for (int f = 0; f < EC1; ++f) {
        const float dy = fmaf(c[f], dr[(size_t)f * g.Np], fmaf(c[2 * EC1 + f], yr[(size_t)f * g.Np], c[EC1 + f]));
        a0 = fmaf(w[3 * f], dy, a0); a1 = fmaf(w[3 * f + 1], dy, a1); a2 = fmaf(w[3 * f + 2], dy, a2);
    }
    float *o = dx + (size_t)b * 3 * g.N + p;
    o[0] = a0; o[g.N] = a1; o[2 * (size_t)g.N] = a2;
}

// ---- host side ------------------------------------------------------------------------------------------------------
Geo make_geo(int B, int N) {
    Geo g;
    g.B = B; g.N = N; g.Np = (N + TILE - 1) / TILE * TILE; g.tpc = g.Np / TILE; g.ptiles = B * g.tpc;
    g.nwg = (g.ptiles + PW - 1) / PW; g.P = (long)B * g.Np;
    return g;
}
// point k-steps per weight-gradient workgroup: 32 (512 points) up to 131 072 points, then as many as keep the number of
// split-K partial blocks (nchunk x 512 KB for dW_3) at 256
inline int ks_chunk_of(long PS) {
    long k = (PS + 255) / 256;
    k = (k + 1) & ~1L;
    return (int)(k < 32 ? 32 : k);
}

struct TWork {
    float *y[4], *dz[3];
    uint8_t *aP[3], *wf[3], *wb[3];
    float *part, *cnt, *bnp, *coef, *yarg, *gz, *wpart, *w0part, *tmax, *tmin;
    int *arg;
};
size_t t_carve(void *ws, const Geo &g, TWork *w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { void *p = ws ? (uint8_t *)ws + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    const int C[4] = {EC1, EC2, EC3, EC4};
    void *p;
    for (int l = 0; l < 4; ++l) { p = take((size_t)C[l] * g.P * 4); if (w) w->y[l] = (float *)p; }
    for (int l = 0; l < 3; ++l) { p = take((size_t)C[l] * g.P * 4); if (w) w->dz[l] = (float *)p; }
    for (int l = 0; l < 3; ++l) { p = take((size_t)C[l] * g.P * 4); if (w) w->aP[l] = (uint8_t *)p; }
    for (int l = 1; l < 4; ++l) {
        p = take((size_t)C[l] * C[l - 1] * 6); if (w) w->wf[l - 1] = (uint8_t *)p;
        p = take((size_t)C[l] * C[l - 1] * 4); if (w) w->wb[l - 1] = (uint8_t *)p;
    }
    p = take((size_t)g.nwg * 2 * EC4 * 4); if (w) w->part = (float *)p;
    p = take((size_t)g.nwg * 4); if (w) w->cnt = (float *)p;
    p = take((size_t)4 * TCSUM * 4); if (w) w->bnp = (float *)p;
    p = take((size_t)3 * TCSUM * 4); if (w) w->coef = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->yarg = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->gz = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->arg = (int *)p;
    const long nchunk = (g.P / 16 + ks_chunk_of(g.P / 16) - 1) / ks_chunk_of(g.P / 16);
    p = take((size_t)nchunk * EC4 * EC3 * 4); if (w) w->wpart = (float *)p;
    p = take((size_t)g.B * EC1 * 3 * 4); if (w) w->w0part = (float *)p;
    p = take((size_t)g.ptiles * EC4 * 4); if (w) w->tmax = (float *)p;
    p = take((size_t)g.ptiles * EC4 * 4); if (w) w->tmin = (float *)p;
    return off;
}

inline const float *cW(const float *canon, int l) { return canon + e_layer_off(l); }
inline const float *cG(const float *canon, int l) { return canon + e_layer_off(l) + e_cout(l) * e_cin(l); }

template <int KS, int NT, int MODE, int NS, bool POOL = false>
int launch_pgemm(PArgs a, hipStream_t s) {
    constexpr int K = KS * 16, NPAR = MODE == FWD ? 2 : 3;
    a.nch = a.Ntot / (NT * 32);
    const int lds = 2 * 2 * NT * NS * 1024 + NPAR * K * 4 + PW * 2 * NT * 32 * 4 + (MODE == BWD_SPARSE ? PW * K * 8 : 0);
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)et_pgemm_kernel<KS, NT, MODE, NS, POOL>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((et_pgemm_kernel<KS, NT, MODE, NS, POOL>), dim3((a.g.nwg + 7) / 8 * 8 * a.nch), dim3(PW * 64), lds, s, a);
    return (int)hipGetLastError();
}
template <int NS>
int forward_gemms(const Geo &g, const TWork &w, hipStream_t s, const int (&C)[4], int l) {
    PArgs a{};
    a.g = g; a.yin = w.y[l - 1]; a.pin = w.bnp + 4 * t_coff(l - 1); a.wpk = w.wf[l - 1]; a.out = w.y[l]; a.part = w.part; a.Ntot = C[l];
    a.tmax = w.tmax; a.tmin = w.tmin;
    if (l == 1) return launch_pgemm<4, 4, FWD, NS>(a, s);
    if (l == 2) return launch_pgemm<8, 4, FWD, NS>(a, s);
    return launch_pgemm<16, 4, FWD, NS, true>(a, s);
}
template <int WM, int WN, int GM, int GN, int MODE>
int launch_wgrad(const WArgs &a, int nchunk, int rblocks, int cblocks, hipStream_t s) {
    const int lds = 2 * 2 * (GM * WM + GN * WN) * 2048;
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)et_wgrad_kernel<WM, WN, GM, GN, MODE>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((et_wgrad_kernel<WM, WN, GM, GN, MODE>), dim3(nchunk, rblocks, cblocks), dim3(GM * GN * 64), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

#define ET_CHECK(expr) do { int e_ = (expr); if (e_) return e_; } while (0)
#define ET_LAST() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" size_t dpf_encoder_train_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return t_carve(nullptr, make_geo(B, N), nullptr);
}

static int encoder_train_forward_direct(int B, int N, int precision, const float *canon, const float *x, void *ws, float *pooled,
                                        float *batch_stats, float *const *running, float momentum, dpf_stream_t stream);
static int encoder_train_backward_direct(int B, int N, const float *canon, const float *x, void *ws, const float *pooled,
                                         const float *g_pooled, float *dcanon, float *dx, dpf_stream_t stream);

// The ~13 (forward) / ~17 (backward) launches of a call as one graph launch once the same call has been seen twice
// (graph_cache.h): in a training loop the allocator hands back the same workspace, so the steady state is all replays.
extern "C" int dpf_encoder_train_forward(int B, int N, int precision, const float *canon, const float *x, void *ws, float *pooled,
                                         float *batch_stats, float *const *running, float momentum, dpf_stream_t stream) {
    auto direct = [&](hipStream_t st) {
        return encoder_train_forward_direct(B, N, precision, canon, x, ws, pooled, batch_stats, running, momentum, (dpf_stream_t)st);
    };
    static GraphCache cache;
    GraphKey k;
    k.val(B); k.val(N); k.val(precision); k.val(canon); k.val(x); k.val(ws); k.val(pooled); k.val(batch_stats); k.val(momentum);
    const int has_running = running != nullptr;
    k.val(has_running);
    if (has_running) k.add(running, sizeof(float *) * 8);
    int dev = 0;
    (void)hipGetDevice(&dev);
    k.val(dev);
    return cache.run(k, (hipStream_t)stream, direct);
}

extern "C" int dpf_encoder_train_backward(int B, int N, const float *canon, const float *x, void *ws, const float *pooled,
                                          const float *g_pooled, float *dcanon, float *dx, dpf_stream_t stream) {
    auto direct = [&](hipStream_t st) {
        return encoder_train_backward_direct(B, N, canon, x, ws, pooled, g_pooled, dcanon, dx, (dpf_stream_t)st);
    };
    static GraphCache cache;
    GraphKey k;
    k.val(B); k.val(N); k.val(canon); k.val(x); k.val(ws); k.val(pooled); k.val(g_pooled); k.val(dcanon); k.val(dx);
    int dev = 0;
    (void)hipGetDevice(&dev);
    k.val(dev);
    return cache.run(k, (hipStream_t)stream, direct);
}

static int encoder_train_forward_direct(int B, int N, int precision, const float *canon, const float *x, void *ws, float *pooled,
                                        float *batch_stats, float *const *running, float momentum, dpf_stream_t stream) {
    if (B <= 0 || N <= 0) return DPF_EINVAL;
    if (!canon || !x || !ws || !pooled) return DPF_EINVAL;
    if ((long)B * N < 2) return DPF_EINVAL;                   // BatchNorm1d refuses a single value per channel in training
    if (precision != DPF_PREC_BF16X3 && precision != DPF_PREC_BF16X6) return DPF_EINVAL;
    const bool x6 = precision == DPF_PREC_BF16X6;
    hipStream_t s = (hipStream_t)stream;
    const Geo g = make_geo(B, N);
    TWork w;
    t_carve(ws, g, &w);
    const double count = (double)B * N;
    const int C[4] = {EC1, EC2, EC3, EC4};
    // weights -> fragments: forward Bm[k = cin][n = cout] = W[n][k]; backward Bm[k = cout][n = cin] = W[k][n]
    WPackArgs wp;
    for (int l = 1; l < 4; ++l) {
        const int cin = C[l - 1], cout = C[l];
        const int ntb = cin >= 128 ? 4 : 2;
        wp.job[l - 1] = WPackJob{cW(canon, l), w.wf[l - 1], 1, cin, cin / 16, 4, cout / 128, x6 ? 3 : 2};
        wp.job[l + 2] = WPackJob{cW(canon, l), w.wb[l - 1], cin, 1, cout / 16, ntb, cin / 32 / ntb, 2};
    }
    hipLaunchKernelGGL(et_wpack_kernel, dim3(32, 6), dim3(256), 0, s, wp);
    hipLaunchKernelGGL(et_l0_kernel, dim3(g.nwg), dim3(256), 0, s, g, cW(canon, 0), x, w.y[0], w.part, w.cnt);
    ET_LAST();
    auto finish = [&](int l) {
        const float *gam = cG(canon, l);
        hipLaunchKernelGGL(et_bn_finish_kernel, dim3(C[l] / FF), dim3(FF * FS), 0, s, g.nwg, C[l], count, w.part, w.cnt, gam, gam + C[l],
                           w.bnp + 4 * t_coff(l), running ? running[2 * l] : nullptr, running ? running[2 * l + 1] : nullptr, momentum,
                           batch_stats ? batch_stats + 2 * t_coff(l) : nullptr);
    };
    finish(0);
    for (int l = 1; l < 4; ++l) {
        ET_CHECK(x6 ? forward_gemms<3>(g, w, s, C, l) : forward_gemms<2>(g, w, s, C, l));
        finish(l);
    }
    hipLaunchKernelGGL(et_pool_kernel, dim3((B * EC4 + 63) / 64), dim3(64), 0, s, g, w.y[3], w.tmax, w.tmin, w.bnp + 4 * t_coff(3), pooled,
                       w.arg, w.yarg);
    return (int)hipGetLastError();
}

static int encoder_train_backward_direct(int B, int N, const float *canon, const float *x, void *ws, const float *pooled,
                                         const float *g_pooled, float *dcanon, float *dx, dpf_stream_t stream) {
    if (B <= 0 || N <= 0) return DPF_EINVAL;
    if (!canon || !x || !ws || !pooled || !g_pooled || !dcanon) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const Geo g = make_geo(B, N);
    TWork w;
    t_carve(ws, g, &w);
    const double count = (double)B * N;
    const int C[4] = {EC1, EC2, EC3, EC4};
    const long PS = g.P / 16;
    const int KS_CHUNK = ks_chunk_of(PS);
    const int nchunk = (int)((PS + KS_CHUNK - 1) / KS_CHUNK);
    auto dG = [&](int l) { return dcanon + e_layer_off(l) + e_cout(l) * e_cin(l); };

    hipLaunchKernelGGL(et_pool_bwd_kernel, dim3(EC4 / 64), dim3(256), 0, s, B, g_pooled, pooled, w.yarg, w.bnp + 4 * t_coff(3), w.gz, w.part);
    ET_LAST();
    for (int l = 3; l >= 1; --l) {
        const int cin = C[l - 1], cout = C[l];
        float *coef = w.coef + 3 * t_coff(l);
        hipLaunchKernelGGL(et_bn_bwd_finish_kernel, dim3(cout / FF), dim3(FF * FS), 0, s, l == 3 ? 1 : g.nwg, cout, count, w.part,
                           w.bnp + 4 * t_coff(l), coef, dG(l), dG(l) + cout);
        // dz_{l-1}, its BatchNorm sums, and a_{l-1} as the other operand of dW_l
        PArgs a{};
        a.g = g; a.yin = w.y[l]; a.dzin = l == 3 ? nullptr : w.dz[l]; a.pin = coef; a.arg = w.arg; a.gz = w.gz; a.wpk = w.wb[l - 1];
        a.out = w.dz[l - 1]; a.yprev = w.y[l - 1]; a.bnprev = w.bnp + 4 * t_coff(l - 1); a.part = w.part; a.Ntot = cin; a.apk = w.aP[l - 1];
        if (l == 3) ET_CHECK((launch_pgemm<32, 4, BWD_SPARSE, 2>(a, s)));
        if (l == 2) ET_CHECK((launch_pgemm<16, 4, BWD, 2>(a, s)));
        if (l == 1) ET_CHECK((launch_pgemm<8, 2, BWD, 2>(a, s)));
        WArgs wa{g, w.y[l], l == 3 ? nullptr : w.dz[l], coef, w.arg, w.gz, w.aP[l - 1], w.wpart, PS, KS_CHUNK, cout, cin};
        if (l == 3) ET_CHECK((launch_wgrad<4, 2, 2, 4, BWD_SPARSE>(wa, nchunk, 2, 1, s)));
        if (l == 2) ET_CHECK((launch_wgrad<2, 2, 2, 2, BWD>(wa, nchunk, 2, 1, s)));
        if (l == 1) ET_CHECK((launch_wgrad<1, 1, 2, 2, BWD>(wa, nchunk, 2, 1, s)));
        const int E4 = cout * cin / 4;
        hipLaunchKernelGGL(et_wreduce_kernel, dim3(E4 / 64), dim3(256), 0, s, nchunk, E4, (const f32x4 *)w.wpart,
                           (f32x4 *)(dcanon + e_layer_off(l)));
    }
    float *coef0 = w.coef;
    hipLaunchKernelGGL(et_bn_bwd_finish_kernel, dim3(EC1 / FF), dim3(FF * FS), 0, s, g.nwg, EC1, count, w.part, w.bnp, coef0, dG(0), dG(0) + EC1);
    hipLaunchKernelGGL(et_wgrad0_kernel, dim3(EC1, B), dim3(256), 0, s, g, w.y[0], w.dz[0], coef0, x, w.w0part);
    hipLaunchKernelGGL(et_w0reduce_kernel, dim3(1), dim3(768), 0, s, B, w.w0part, dcanon);
    if (dx != nullptr)
        hipLaunchKernelGGL(et_dx_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, g, w.y[0], w.dz[0], coef0, cW(canon, 0), dx);
    return (int)hipGetLastError();
}
