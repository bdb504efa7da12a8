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
// live in HBM: at cfg-2 that is 252 MB, a few percent of the part, read and written at 4-5 TB/s.
//
//  forward   et_wpack     W_l as bf16 hi/lo MFMA B-operand fragments, both orientations (forward and W^T)
//            et_l0        y0 = W0 x (fp32 FMA), per-workgroup sum / sum of squares
//            et_bn_finish per-workgroup partials -> mean, 1/std, folded scale/shift; running statistics
//            et_pgemm<FWD> y_l = W_l relu(BN(y_{l-1})): the BatchNorm + ReLU + bf16 hi/lo split is the PROLOGUE on the
//                         operand registers (the activations a_l are never stored); epilogue stores y_l and emits the
//                         per-workgroup sum / sum of squares                                       (l = 1, 2, 3)
//            et_pool      pooled[b, f] = relu(max_p BN(y3)), the argmax and y3 there
//  backward  et_pool_bwd  the pooled gradient is a sparse d z3 (one point per (b, f)); its BatchNorm sums
//            et_bn_bwd_finish  d gamma, d beta and the coefficients of  d y_l = c0 dz_l + c1 + c2 y_l
//            et_packp     d y_l and a_{l-1} as bf16 hi/lo fragments with K = points (operands of the weight gradient)
//            et_pgemm<BWD> G = W_l^T d y_l (prologue forms d y_l from dz_l and y_l); epilogue masks with the ReLU of
//                         layer l-1, stores dz_{l-1} and emits its BatchNorm sums
//            et_wgrad     dW_l = d y_l a_{l-1}^T, split over the points; et_wreduce adds the splits in a fixed order
//            et_wgrad0    dW_0 = d y_0 x^T (fp32)
//
// The forward contractions -- which decide every ReLU mask and the argmax -- run at the precision the caller asks for:
// bf16x6 (hi/mid/lo split, six products, fp32-class; the host side's default: a ReLU whose pre-activation is within
// the forward error of zero takes the other subgradient than the reference's, and at 1e-5 that happens to ~1e-5 of
// all ReLUs) or bf16x3.  The gradient contractions use hi/lo splits (hi*hi + hi*lo + lo*hi), fp32 accumulation.
// Every sum is taken in a fixed order (no floating-point atomics): the result is deterministic.
// d(input) is not produced (the clouds are data); the host side keeps the tensor-op path for a differentiable input.
#include "flow_common.h"
#include "encoder_layout.h"

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
// n = 32 (chunk NT + nt) + (lane & 31), Bm[k][n] = W[k * sk + n * sn]
template <int NS>
__global__ __launch_bounds__(256) void et_wpack_kernel(const float *__restrict__ W, int sk, int sn, int KS, int NT, int nchunk,
                                                        uint8_t *__restrict__ out) {
    const int total = nchunk * KS * NT * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, frag = idx >> 6;
        const int nt = frag % NT, ks = (frag / NT) % KS, chunk = frag / (NT * KS);
        const int n = 32 * (chunk * NT + nt) + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W[(size_t)(k0 + j) * sk + (size_t)n * sn];
        u32x4 part[NS];
        split8<NS>(v, part);
        uint8_t *o = out + ((size_t)frag * NS) * 1024 + lane * 16;
#pragma unroll
        for (int q = 0; q < NS; ++q) *(u32x4 *)(o + q * 1024) = part[q];
    }
}

// ---- layer 0: y0 = W0 x --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void et_l0_kernel(Geo g, const float *__restrict__ W0, const float *__restrict__ x,
                                                     float *__restrict__ y0, float *__restrict__ part) {
    __shared__ float w[EC1 * EC0];
    __shared__ float red[4][2][EC1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < EC1 * EC0) w[tid] = W0[tid];
    __syncthreads();
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
    for (int f = 0; f < EC1; ++f) {
        const float v = live ? fmaf(w[3 * f + 2], x2, fmaf(w[3 * f + 1], x1, w[3 * f] * x0)) : 0.f;
        if (tile_ok) yo[(size_t)f * g.Np] = v;
        const float s1 = wave_sum(v), s2 = wave_sum(v * v);
        if (lane == 0) { red[wave][0][f] = s1; red[wave][1][f] = s2; }
    }
    __syncthreads();
    if (tid < 2 * EC1) {
        const int which = tid >> 6, f = tid & 63;
        part[((size_t)blockIdx.x * 2 + which) * EC1 + f] = (red[0][which][f] + red[1][which][f]) + (red[2][which][f] + red[3][which][f]);
    }
}

// ---- BatchNorm statistics: partials -> folded parameters -----------------------------------------------------------
// bnp: [4][C] = scale (gamma / std), shift (beta - mean * scale), mean, 1 / std
__global__ __launch_bounds__(64) void et_bn_finish_kernel(int nwg, int C, double count, const float *__restrict__ part,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           float *__restrict__ bnp, float *__restrict__ run_mean,
                                                           float *__restrict__ run_var, float momentum, float *__restrict__ bstat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int w = 0; w < nwg; ++w) {
        s1 += (double)part[((size_t)w * 2) * C + c];
        s2 += (double)part[((size_t)w * 2 + 1) * C + c];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
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
__global__ __launch_bounds__(64) void et_bn_bwd_finish_kernel(int nwg, int C, double count, const float *__restrict__ part,
                                                               const float *__restrict__ bnp, float *__restrict__ coef,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < nwg; ++w) {
        t1 += (double)part[((size_t)w * 2) * C + c];
        t2 += (double)part[((size_t)w * 2 + 1) * C + c];
    }
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
enum { FWD = 0, BWD = 1, BWD_SPARSE = 2 };
struct PArgs {
    Geo g;
    const float *yin;       // (B, K, Np): FWD y_{l-1}; BWD y_l
    const float *dzin;      // (B, K, Np): BWD dz_l
    const float *pin;       // FWD: bnp of layer l-1 [4][K]; BWD: coef of layer l [3][K]
    const int *arg;         // BWD_SPARSE: (B, K) argmax point
    const float *gz;        // BWD_SPARSE: (B, K) masked pooled gradient
    const uint8_t *wpk;     // [nchunk][KS][NT][NS][64][16 B]
    float *out;             // (B, Ntot, Np): FWD y_l; BWD dz_{l-1}
    const float *yprev;     // BWD: (B, Ntot, Np) y_{l-1}
    const float *bnprev;    // BWD: bnp of layer l-1 [4][Ntot]
    float *part;            // [nwg][2][Ntot]
    int Ntot;
};

template <int KS, int NT, int MODE, int NS>
__global__ __launch_bounds__(PW * 64) void et_pgemm_kernel(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef Terms<NS> TT;
    constexpr int K = KS * 16, KC = 16 / NT, NCH = KS / KC, BUF = KC * NT * NS * 1024;   // 16 KiB of fragments per part and buffer
    constexpr int NG = NT < 4 ? NT : 4;                      // output tiles whose MFMAs are interleaved
    static_assert(KS % KC == 0 && BUF == NS * 16384, "chunking");
    constexpr int NPAR = MODE == FWD ? 2 : 3;
    uint8_t *l_w = smem;                                     // [2][BUF]
    float *l_par = (float *)(smem + 2 * BUF);                // [NPAR][K]
    float *l_red = l_par + NPAR * K;                         // [PW][2][NT * 32]

    const Geo &g = a.g;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.y;
    int ptile = blockIdx.x * PW + wave;
    const bool tile_ok = ptile < g.ptiles;
    if (!tile_ok) ptile = g.ptiles - 1;
    const int b = ptile / g.tpc, p0 = (ptile % g.tpc) * TILE;
    const bool live = tile_ok && p0 + pl < g.N;              // this lane's point exists

    const uint8_t *wsrc = a.wpk + (size_t)chunk * KS * NT * NS * 1024;
    auto stage = [&](int c) {                                // chunk c of the weight stream -> buffer c & 1
        const uint8_t *src = wsrc + (size_t)c * BUF;
        uint8_t *dst = l_w + (c & 1) * BUF;
#pragma unroll
        for (int i = 0; i < BUF / 1024 / PW; ++i) {
            const int k = wave + i * PW;
            __builtin_amdgcn_global_load_lds((glb_void *)(src + k * 1024 + lane * 16), (lds_void *)(dst + k * 1024), 16, 0, 0);
        }
    };
    stage(0);
    if (NCH > 1) stage(1);
    for (int i = threadIdx.x; i < NPAR * K; i += PW * 64) l_par[i] = a.pin[i];

    // operand rows of this lane: features 16 s + 8 h + j of point p0 + pl
    const size_t in_base = (size_t)b * K * g.Np + p0 + pl;
    float ry[8], rz[8];
    auto fetch = [&](int s) {
        const size_t o = in_base + (size_t)(16 * s + 8 * h) * g.Np;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ry[j] = a.yin[o + (size_t)j * g.Np];
            if (MODE == BWD) rz[j] = a.dzin[o + (size_t)j * g.Np];
        }
        if (MODE == BWD_SPARSE) {
            const int f0 = 16 * s + 8 * h;
            const int4 a0 = *(const int4 *)(a.arg + (size_t)b * K + f0), a1 = *(const int4 *)(a.arg + (size_t)b * K + f0 + 4);
            const f32x4 g0 = *(const f32x4 *)(a.gz + (size_t)b * K + f0), g1 = *(const f32x4 *)(a.gz + (size_t)b * K + f0 + 4);
            const int p = p0 + pl;
            rz[0] = a0.x == p ? g0.x : 0.f; rz[1] = a0.y == p ? g0.y : 0.f; rz[2] = a0.z == p ? g0.z : 0.f; rz[3] = a0.w == p ? g0.w : 0.f;
            rz[4] = a1.x == p ? g1.x : 0.f; rz[5] = a1.y == p ? g1.y : 0.f; rz[6] = a1.z == p ? g1.z : 0.f; rz[7] = a1.w == p ? g1.w : 0.f;
        }
    };
    auto operand = [&](int s, u32x4 (&frag)[NS]) {           // prologue: raw rows -> the NS bf16 parts of the A fragment
        const int f0 = 16 * s + 8 * h;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 c0 = *(const f32x4 *)(l_par + f0 + 4 * q), c1 = *(const f32x4 *)(l_par + K + f0 + 4 * q);
            f32x4 c2 = c1;
            if (MODE != FWD) c2 = *(const f32x4 *)(l_par + 2 * K + f0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;
                float r;
                if (MODE == FWD) r = fmaxf(fmaf(ry[j], c0[e], c1[e]), 0.f);
                else r = fmaf(c0[e], rz[j], fmaf(c2[e], ry[j], c1[e]));
                v[j] = live ? r : 0.f;
            }
        }
        split8<NS>(v, frag);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();                                     // chunk c has landed (and l_par on the first pass)
        const uint8_t *wb = l_w + (c & 1) * BUF;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int s = c * KC + kc;
            u32x4 af[NS];
            operand(s, af);
            if (s + 1 < KS) fetch(s + 1);
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
                    for (int i = 0; i < NG; ++i) acc[ng + i] = mfma(af[TT::A[term]], wf[i][TT::B[term]], acc[ng + i]);
            }
        }
        if (c + 2 < NCH) {
            __syncthreads();                                 // everybody is done with buffer c & 1
            stage(c + 2);
        }
    }

    // epilogue: accumulator register r = point p0 + (r & 3) + 8 (r >> 2) + 4 h, lane column = feature n0 + 32 nt + pl
    const int n0 = chunk * NT * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int f = n0 + 32 * nt + pl;
        const size_t o = ((size_t)b * a.Ntot + f) * g.Np + p0 + 4 * h;
        float s1 = 0.f, s2 = 0.f;
        float sc = 0.f, sh = 0.f, mean = 0.f, rstd = 0.f;
        if (MODE != FWD) { sc = a.bnprev[f]; sh = a.bnprev[a.Ntot + f]; mean = a.bnprev[2 * a.Ntot + f]; rstd = a.bnprev[3 * a.Ntot + f]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
            if (MODE == FWD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
            } else {
                const f32x4 y = *(const f32x4 *)(a.yprev + o + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(y[e], sc, sh) > 0.f ? v[e] : 0.f;
                    s1 += v[e];
                    s2 = fmaf(v[e], (y[e] - mean) * rstd, s2);
                }
            }
            if (tile_ok) *(f32x4 *)(a.out + o + 8 * q) = v;
        }
        s1 = half_add(s1); s2 = half_add(s2);
        if (!tile_ok) { s1 = 0.f; s2 = 0.f; }
        if (!h) {
            l_red[(wave * 2 + 0) * (NT * 32) + 32 * nt + pl] = s1;
            l_red[(wave * 2 + 1) * (NT * 32) + 32 * nt + pl] = s2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NT * 32; i += PW * 64) {
        const int which = i / (NT * 32), f = i % (NT * 32);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < PW; ++w) s += l_red[(w * 2 + which) * (NT * 32) + f];
        a.part[((size_t)blockIdx.x * 2 + which) * a.Ntot + n0 + f] = s;
    }
}

// ---- max over the points -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void et_pool_kernel(Geo g, const float *__restrict__ y3, const float *__restrict__ bnp,
                                                       float *__restrict__ pooled, int *__restrict__ arg, float *__restrict__ yarg) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= g.B * EC4) return;
    const int f = row % EC4;
    const float sc = bnp[f], sh = bnp[EC4 + f];
    const float *yr = y3 + (size_t)row * g.Np;
    float best = -__builtin_inff(), by = 0.f;
    int bi = 0x7fffffff;
    for (int p = lane * 4; p < g.N; p += 256) {
        const f32x4 y = *(const f32x4 *)(yr + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(y[e], sc, sh);
            if (p + e < g.N && z > best) { best = z; by = y[e]; bi = p + e; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), oy = __shfl_xor(by, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; by = oy; bi = oi; }
    }
    if (lane == 0) {
        pooled[row] = fmaxf(best, 0.f);
        arg[row] = bi;
        yarg[row] = by;
    }
}

// the pooled gradient as the sparse dz3: gz[b, f] = g[b, f] [pooled > 0] at point arg[b, f]; its BatchNorm sums
__global__ __launch_bounds__(64) void et_pool_bwd_kernel(int B, const float *__restrict__ gp, const float *__restrict__ pooled,
                                                          const float *__restrict__ yarg, const float *__restrict__ bnp,
                                                          float *__restrict__ gz, float *__restrict__ part) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= EC4) return;
    const float mean = bnp[2 * EC4 + f], rstd = bnp[3 * EC4 + f];
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float v = pooled[b * EC4 + f] > 0.f ? gp[b * EC4 + f] : 0.f;
        gz[b * EC4 + f] = v;
        s1 += v;
        s2 = fmaf(v, (yarg[b * EC4 + f] - mean) * rstd, s2);
    }
    part[f] = s1;
    part[EC4 + f] = s2;
}

// ---- operands of the weight gradient: rows = features, K = points ----------------------------------------------------
// out: [ftile C/32][kstep P/16][part 2][lane 64][8 bf16]; lane (row, kg) holds points 16 s + 8 kg + j of feature 32 ft + row.
// MODE FWD: a = relu(BN(y)); BWD: d y = c0 dz + c1 + c2 y; BWD_SPARSE: dz = gz at the argmax point.  Padded points give 0.
template <int MODE>
__global__ __launch_bounds__(256) void et_packp_kernel(Geo g, int C, const float *__restrict__ y, const float *__restrict__ dz,
                                                        const float *__restrict__ par, const int *__restrict__ arg,
                                                        const float *__restrict__ gz, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, row = lane & 31, kg = lane >> 5;
    const long PS = g.P / 16, items = (long)(C / 32) * (PS / 2);
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= items) return;
    const int ft = (int)(item / (PS / 2));
    const long s2 = item % (PS / 2);
    const int f = 32 * ft + row;
    const float c0 = par[f], c1 = par[C + f], c2 = MODE == FWD ? 0.f : par[2 * C + f];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const long s = 2 * s2 + sp, q = 16 * s + 8 * kg;
        const int b = (int)(q / g.Np), p = (int)(q % g.Np);
        const size_t o = ((size_t)b * C + f) * g.Np + p;
        const f32x4 y0 = *(const f32x4 *)(y + o), y1 = *(const f32x4 *)(y + o + 4);
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        float v[8];
        if (MODE == FWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(yy[j], c0, c1), 0.f);
        } else if (MODE == BWD) {
            const f32x4 d0 = *(const f32x4 *)(dz + o), d1 = *(const f32x4 *)(dz + o + 4);
            const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(c0, dd[j], fmaf(c2, yy[j], c1));
        } else {
            const int ap = arg[(size_t)b * C + f];
            const float gv = gz[(size_t)b * C + f];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(c0, ap == p + j ? gv : 0.f, fmaf(c2, yy[j], c1));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (p + j >= g.N) v[j] = 0.f;
        u32x4 hl[2];
        split8<2>(v, hl);
        uint8_t *dst = out + (((size_t)ft * PS + s) * 2) * 1024 + lane * 16;
        *(u32x4 *)dst = hl[0];
        *(u32x4 *)(dst + 1024) = hl[1];
    }
}

// ---- weight gradient: part[kchunk][m][n] = sum over the chunk's points of A[m][p] Bm[n][p] ---------------------------------
struct WArgs {
    const uint8_t *A, *Bm;      // packed K = points operands: [tile][PS][2][64][16 B]
    float *part;                // [nchunk][M][Ncols]
    long PS;
    int ks_chunk, M, Ncols;
};

template <int WM, int WN, int GM, int GN>
__global__ __launch_bounds__(GM * GN * 64) void et_wgrad_kernel(WArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NWV = GM * GN, BM = GM * WM, BN = GN * WN, KC = 2;
    constexpr int STAGE = KC * (BM + BN) * 2048, NI = STAGE / 1024;
    static_assert(NI % NWV == 0, "staging");
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / GN, wn = wave % GN;
    const long s_begin = (long)blockIdx.x * a.ks_chunk;
    const long s_end_l = s_begin + a.ks_chunk < a.PS ? s_begin + a.ks_chunk : a.PS;
    const int nst = (int)((s_end_l - s_begin + KC - 1) / KC);
    const int rb = blockIdx.y, cb = blockIdx.z;

    auto stage = [&](int st) {      // k-steps s_begin + st KC .. of all block tiles -> buffer st & 1: [kc][tile BM + BN][part]
        uint8_t *dst = smem + (st & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < NI / NWV; ++i) {
            const int k = wave + i * NWV;
            const int part = k & 1, tile = (k >> 1) % (BM + BN), kc = (k >> 1) / (BM + BN);
            long s = s_begin + (long)st * KC + kc;
            if (s >= a.PS) s = a.PS - 1;                      // tail: a repeated k-step, not used
            const uint8_t *src = tile < BM ? a.A + ((((size_t)(rb * BM + tile)) * a.PS + s) * 2 + part) * 1024
                                           : a.Bm + ((((size_t)(cb * BN + tile - BM)) * a.PS + s) * 2 + part) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void *)(src + lane * 16), (lds_void *)(dst + k * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    stage(0);
    if (nst > 1) stage(1);
    for (int st = 0; st < nst; ++st) {
        __syncthreads();
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
        if (st + 2 < nst) {
            __syncthreads();
            stage(st + 2);
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

// out[e] = sum over the chunks of part[chunk][e], fixed order
__global__ __launch_bounds__(256) void et_wreduce_kernel(int nchunk, int E4, const f32x4 *__restrict__ part, f32x4 *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E4) return;
    f32x4 s = part[e];
    for (int c = 1; c < nchunk; ++c) {
        const f32x4 v = part[(size_t)c * E4 + e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[e] = s;
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
__global__ __launch_bounds__(192) void et_w0reduce_kernel(int B, const float *__restrict__ part, float *__restrict__ out) {
    const int e = threadIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(size_t)b * EC1 * 3 + e];
    out[e] = s;
}

// ---- host side ------------------------------------------------------------------------------------------------------
Geo make_geo(int B, int N) {
    Geo g;
    g.B = B; g.N = N; g.Np = (N + TILE - 1) / TILE * TILE; g.tpc = g.Np / TILE; g.ptiles = B * g.tpc;
    g.nwg = (g.ptiles + PW - 1) / PW; g.P = (long)B * g.Np;
    return g;
}
constexpr int KS_CHUNK = 32;                 // point k-steps (512 points) per weight-gradient workgroup

struct TWork {
    float *y[4], *dz[3];
    uint8_t *aP[3], *dyP, *wf[3], *wb[3];
    float *part, *bnp, *coef, *yarg, *gz, *wpart, *w0part;
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
    p = take((size_t)EC4 * g.P * 4); if (w) w->dyP = (uint8_t *)p;
    for (int l = 1; l < 4; ++l) {
        p = take((size_t)C[l] * C[l - 1] * 6); if (w) w->wf[l - 1] = (uint8_t *)p;
        p = take((size_t)C[l] * C[l - 1] * 4); if (w) w->wb[l - 1] = (uint8_t *)p;
    }
    p = take((size_t)g.nwg * 2 * EC4 * 4); if (w) w->part = (float *)p;
    p = take((size_t)4 * TCSUM * 4); if (w) w->bnp = (float *)p;
    p = take((size_t)3 * TCSUM * 4); if (w) w->coef = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->yarg = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->gz = (float *)p;
    p = take((size_t)g.B * EC4 * 4); if (w) w->arg = (int *)p;
    const long nchunk = (g.P / 16 + KS_CHUNK - 1) / KS_CHUNK;
    p = take((size_t)nchunk * EC4 * EC3 * 4); if (w) w->wpart = (float *)p;
    p = take((size_t)g.B * EC1 * 3 * 4); if (w) w->w0part = (float *)p;
    return off;
}

inline const float *cW(const float *canon, int l) { return canon + e_layer_off(l); }
inline const float *cG(const float *canon, int l) { return canon + e_layer_off(l) + e_cout(l) * e_cin(l); }

template <int KS, int NT, int MODE, int NS>
int launch_pgemm(const PArgs &a, int nchunk, hipStream_t s) {
    constexpr int K = KS * 16, NPAR = MODE == FWD ? 2 : 3;
    const int lds = 2 * NS * 16384 + NPAR * K * 4 + PW * 2 * NT * 32 * 4;
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)et_pgemm_kernel<KS, NT, MODE, NS>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((et_pgemm_kernel<KS, NT, MODE, NS>), dim3(a.g.nwg, nchunk), dim3(PW * 64), lds, s, a);
    return (int)hipGetLastError();
}
template <int NS>
int forward_layers(const Geo &g, const TWork &w, const float *canon, hipStream_t s, const int (&C)[4]) {
    for (int l = 1; l < 4; ++l) {
        const int cin = C[l - 1], cout = C[l];
        const int ntf = cout >= 256 ? 8 : 4;
        hipLaunchKernelGGL(et_wpack_kernel<NS>, dim3(64), dim3(256), 0, s, cW(canon, l), 1, cin, cin / 16, ntf, cout / 32 / ntf, w.wf[l - 1]);
    }
    return (int)hipGetLastError();
}
template <int WM, int WN, int GM, int GN>
int launch_wgrad(const WArgs &a, int nchunk, int rblocks, int cblocks, hipStream_t s) {
    const int lds = 2 * 2 * (GM * WM + GN * WN) * 2048;
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)et_wgrad_kernel<WM, WN, GM, GN>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((et_wgrad_kernel<WM, WN, GM, GN>), dim3(nchunk, rblocks, cblocks), dim3(GM * GN * 64), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

#define ET_CHECK(expr) do { int e_ = (expr); if (e_) return e_; } while (0)
#define ET_LAST() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" size_t dpf_encoder_train_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return t_carve(nullptr, make_geo(B, N), nullptr);
}

extern "C" int dpf_encoder_train_forward(int B, int N, int precision, const float *canon, const float *x, void *ws, float *pooled,
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
    ET_CHECK(x6 ? forward_layers<3>(g, w, canon, s, C) : forward_layers<2>(g, w, canon, s, C));
    for (int l = 1; l < 4; ++l) {
        const int cin = C[l - 1], cout = C[l];
        const int ntb = l == 3 ? 8 : l == 2 ? 4 : 2;
        hipLaunchKernelGGL(et_wpack_kernel<2>, dim3(64), dim3(256), 0, s, cW(canon, l), cin, 1, cout / 16, ntb, cin / 32 / ntb, w.wb[l - 1]);
    }
    hipLaunchKernelGGL(et_l0_kernel, dim3(g.nwg), dim3(256), 0, s, g, cW(canon, 0), x, w.y[0], w.part);
    ET_LAST();
    auto finish = [&](int l) {
        const float *gam = cG(canon, l);
        hipLaunchKernelGGL(et_bn_finish_kernel, dim3((C[l] + 63) / 64), dim3(64), 0, s, g.nwg, C[l], count, w.part, gam, gam + C[l],
                           w.bnp + 4 * t_coff(l), running ? running[2 * l] : nullptr, running ? running[2 * l + 1] : nullptr, momentum,
                           batch_stats ? batch_stats + 2 * t_coff(l) : nullptr);
    };
    finish(0);
    for (int l = 1; l < 4; ++l) {
        PArgs a{};
        a.g = g; a.yin = w.y[l - 1]; a.pin = w.bnp + 4 * t_coff(l - 1); a.wpk = w.wf[l - 1]; a.out = w.y[l]; a.part = w.part; a.Ntot = C[l];
        if (l == 1) ET_CHECK(x6 ? (launch_pgemm<4, 4, FWD, 3>(a, 1, s)) : (launch_pgemm<4, 4, FWD, 2>(a, 1, s)));
        if (l == 2) ET_CHECK(x6 ? (launch_pgemm<8, 8, FWD, 3>(a, 1, s)) : (launch_pgemm<8, 8, FWD, 2>(a, 1, s)));
        if (l == 3) ET_CHECK(x6 ? (launch_pgemm<16, 8, FWD, 3>(a, 2, s)) : (launch_pgemm<16, 8, FWD, 2>(a, 2, s)));
        finish(l);
    }
    hipLaunchKernelGGL(et_pool_kernel, dim3((B * EC4 + 3) / 4), dim3(256), 0, s, g, w.y[3], w.bnp + 4 * t_coff(3), pooled, w.arg, w.yarg);
    return (int)hipGetLastError();
}

extern "C" int dpf_encoder_train_backward(int B, int N, const float *canon, const float *x, void *ws, const float *pooled,
                                          const float *g_pooled, float *dcanon, dpf_stream_t stream) {
    if (B <= 0 || N <= 0) return DPF_EINVAL;
    if (!canon || !x || !ws || !pooled || !g_pooled || !dcanon) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const Geo g = make_geo(B, N);
    TWork w;
    t_carve(ws, g, &w);
    const double count = (double)B * N;
    const int C[4] = {EC1, EC2, EC3, EC4};
    const long PS = g.P / 16;
    const int nchunk = (int)((PS + KS_CHUNK - 1) / KS_CHUNK);
    auto dG = [&](int l) { return dcanon + e_layer_off(l) + e_cout(l) * e_cin(l); };

    hipLaunchKernelGGL(et_pool_bwd_kernel, dim3(EC4 / 64), dim3(64), 0, s, B, g_pooled, pooled, w.yarg, w.bnp + 4 * t_coff(3), w.gz, w.part);
    ET_LAST();
    for (int l = 3; l >= 1; --l) {
        const int cin = C[l - 1], cout = C[l];
        float *coef = w.coef + 3 * t_coff(l);
        hipLaunchKernelGGL(et_bn_bwd_finish_kernel, dim3(cout / 64), dim3(64), 0, s, l == 3 ? 1 : g.nwg, cout, count, w.part,
                           w.bnp + 4 * t_coff(l), coef, dG(l), dG(l) + cout);
        // operands of dW_l
        const long items_dy = (long)(cout / 32) * (PS / 2), items_a = (long)(cin / 32) * (PS / 2);
        if (l == 3)
            hipLaunchKernelGGL(et_packp_kernel<BWD_SPARSE>, dim3((unsigned)((items_dy + 3) / 4)), dim3(256), 0, s, g, cout, w.y[3], nullptr, coef,
                               w.arg, w.gz, w.dyP);
        else
            hipLaunchKernelGGL(et_packp_kernel<BWD>, dim3((unsigned)((items_dy + 3) / 4)), dim3(256), 0, s, g, cout, w.y[l], w.dz[l], coef,
                               nullptr, nullptr, w.dyP);
        hipLaunchKernelGGL(et_packp_kernel<FWD>, dim3((unsigned)((items_a + 3) / 4)), dim3(256), 0, s, g, cin, w.y[l - 1], nullptr,
                           w.bnp + 4 * t_coff(l - 1), nullptr, nullptr, w.aP[l - 1]);
        ET_LAST();
        WArgs wa{w.dyP, w.aP[l - 1], w.wpart, PS, KS_CHUNK, cout, cin};
        if (l == 3) ET_CHECK((launch_wgrad<4, 2, 2, 4>(wa, nchunk, 2, 1, s)));
        if (l == 2) ET_CHECK((launch_wgrad<2, 2, 4, 2>(wa, nchunk, 1, 1, s)));
        if (l == 1) ET_CHECK((launch_wgrad<1, 1, 4, 2>(wa, nchunk, 1, 1, s)));
        const int E4 = cout * cin / 4;
        hipLaunchKernelGGL(et_wreduce_kernel, dim3((E4 + 255) / 256), dim3(256), 0, s, nchunk, E4, (const f32x4 *)w.wpart,
                           (f32x4 *)(dcanon + e_layer_off(l)));
        // dz_{l-1} and its BatchNorm sums
        PArgs a{};
        a.g = g; a.yin = w.y[l]; a.dzin = l == 3 ? nullptr : w.dz[l]; a.pin = coef; a.arg = w.arg; a.gz = w.gz; a.wpk = w.wb[l - 1];
        a.out = w.dz[l - 1]; a.yprev = w.y[l - 1]; a.bnprev = w.bnp + 4 * t_coff(l - 1); a.part = w.part; a.Ntot = cin;
        if (l == 3) ET_CHECK((launch_pgemm<32, 8, BWD_SPARSE, 2>(a, 1, s)));
        if (l == 2) ET_CHECK((launch_pgemm<16, 4, BWD, 2>(a, 1, s)));
        if (l == 1) ET_CHECK((launch_pgemm<8, 2, BWD, 2>(a, 1, s)));
    }
    float *coef0 = w.coef;
    hipLaunchKernelGGL(et_bn_bwd_finish_kernel, dim3(1), dim3(64), 0, s, g.nwg, EC1, count, w.part, w.bnp, coef0, dG(0), dG(0) + EC1);
    hipLaunchKernelGGL(et_wgrad0_kernel, dim3(EC1, B), dim3(256), 0, s, g, w.y[0], w.dz[0], coef0, x, w.w0part);
    hipLaunchKernelGGL(et_w0reduce_kernel, dim3(1), dim3(192), 0, s, B, w.w0part, dcanon);
    return (int)hipGetLastError();
}
