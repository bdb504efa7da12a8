// Training-mode conditional affine-coupling layer for gfx950 (MI355X): forward with
// batch-statistics BatchNorm and the full backward pass, one coupling layer per call.
//
// Replaces, for model.train(), CondRealNVPFlow3D.forward (lib/networks/flows.py:95-117) and
// what autograd derives from it (lib/networks/training.py:55).  In training mode the layer is
// NOT a per-point map: both BatchNorm1d layers of each conditioner branch normalise with
// statistics over all B*N points (flows.py:27,30,62,65), which puts two grid-wide reductions in
// the forward pass and three in the backward pass.  Each reduction is a kernel boundary:
//
//  forward   tstats_x      moments of the (<=2) kept coordinates       -> BN0 batch stats are an
//            tbn0          analytic function of them (h0 = W0 x is linear): folded input-MFMA
//                          fragments with the BATCH statistics
//            tstats_h1     h1 = W1 relu(BN0(W0 x)) on the matrix cores, sum / sum of squares
//            tfilm_fold    BN1 batch stats + this step's FiLM vectors  -> the per-cloud block the
//                          eval kernel consumes
//            flow_kernel   (csrc/flow.hip, L = 1) the layer itself
//  backward  tbwd1         recompute to h2; d(out) -> dW2, db2, d FiLM(a, c), dh2a (stored)
//            tbwd2         recompute; BN1 backward; dh0 = W1^T dh1 and dW1 = dh1 h0^T on the
//                          matrix cores; dh0a (stored); d gamma0, d beta0
//            tbwd3         BN0 backward; dW0; d(input points)
//            treduce       per-workgroup partial sums -> gradients (deterministic, no atomics)
//
// Activations are RECOMPUTED from the layer input in every backward pass (MFMA work is cheap);
// only the two (B*N, 128) gradient fragments that cross a grid-wide reduction are stored, as raw
// accumulator-fragment dumps.  The per-cloud FiLM conditioner nets (B x 64 tensors) stay on
// PyTorch-ROCm, batched over all layers; they enter here as the tensor `fm` and leave as `dfm`.
//
// Precision: the forward contraction h1 = W1 relu(h0) -- and its recomputation in the backward passes,
// which decides every ReLU mask -- runs at the precision the caller asks for: bf16x3 (hi/lo split,
// ~1e-5) or bf16x6 (hi/mid/lo, fp32-class; the default of the host side, because a ReLU whose
// pre-activation is within the forward error of zero switches the other way and moves that point's
// gradient to the other subgradient: at 1e-5 that happens to ~1e-5 of all ReLUs).  The gradient
// contractions themselves (dh0 = W1^T dh1, dW1 = dh1 h0^T) use hi/lo splits (3 products).
#include <stdlib.h>

#include "flow_common.h"

namespace {

constexpr int TW = 8;                    // waves per workgroup (256 points of one cloud)
constexpr int TBLK = TW * TILE;

// ---- compact per-layer parameter block `tcanon` (floats), branch order (logvar, mu) ----------
constexpr int T_W0 = 0;        // [64][2]  sd0.weight, columns = keep channels (zero column if one)
constexpr int T_G0 = 128;      // [64]     sd0_bn.weight
constexpr int T_B0 = 192;      // [64]     sd0_bn.bias
constexpr int T_W1 = 256;      // [64][64] sd1.weight
constexpr int T_W2 = 4352;     // [2][64]  sd2.weight, rows = warp channels (zero row if one)
constexpr int T_B2 = 4480;     // [4]      sd2.bias
constexpr int T_BR = 4484;
constexpr int T_LAYER = 2 * T_BR;

// ---- packed per-layer training block (bytes); its head is the eval layer format of precision NS ---
constexpr int PT_A1 = 0;                                                   // W1 fragments, NS parts x 16384
__host__ __device__ constexpr int pt_a0(int NS) { return NS * P_A1_PART; }           // input MFMA, gamma*rstd0 / beta
__host__ __device__ constexpr int pt_a0n(int NS) { return pt_a0(NS) + 4096; }        // input MFMA -> NORMALISED h0
__host__ __device__ constexpr int pt_a1t(int NS) { return pt_a0(NS) + 8192; }        // W1^T fragments, hi | lo
__host__ __device__ constexpr int pt_bytes(int NS) { return pt_a1t(NS) + 2 * P_A1_PART; }

// ---- per-layer saved statistics (floats) ---------------------------------------------------------
// stats[br][k][64]: k = 0 mean0, 1 rstd0, 2 mean1, 3 rstd1, 4 batch var0 (unbiased), 5 batch var1 (unbiased)
constexpr int ST_BR = 6 * 64;
constexpr int ST_LAYER = 2 * ST_BR;

// ---- backward FiLM block per (layer, cloud) (floats): [br][k][64], k = 0 a, 1 c, 2 rstd1, 3 c/a ---
constexpr int FB_BR = 4 * 64;
constexpr int FB_CLOUD = 2 * FB_BR;

__device__ __forceinline__ f32x16 zero16() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}

// ===================================================================================================
// pack: W1 and W1^T fragments for every layer (once per optimizer step)
// ===================================================================================================
template <int NS>
__global__ __launch_bounds__(256) void tpack_kernel(const float *__restrict__ tcanon, uint8_t *__restrict__ packed) {
    const int l = blockIdx.x;
    const float *cl = tcanon + (size_t)l * T_LAYER;
    uint16_t *o1 = (uint16_t *)(packed + (size_t)l * pt_bytes(NS) + PT_A1);
    uint16_t *oT = (uint16_t *)(packed + (size_t)l * pt_bytes(NS) + pt_a1t(NS));
    for (int idx = threadIdx.x; idx < 2 * 2 * 4 * 64 * 8; idx += blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 3, tp = (idx >> 11) & 1, br = idx >> 12;
        const int i = lane & 31, h = lane >> 5;
        const int fk = acc_feature(s >> 1, 8 * (s & 1) + j, h);     // feature carried by K slot (s, j, h)
        const float *W1 = cl + br * T_BR + T_W1;
        float r1, r2;
        const float w = W1[(32 * tp + i) * 64 + fk];                // forward: rows = out feature, K = in feature
        o1[idx] = (uint16_t)(split_hi(w, r1) >> 16);
        if (NS == 2) {
            o1[P_A1_PART / 2 + idx] = (uint16_t)bf16_rne(r1);
        } else {
            o1[P_A1_PART / 2 + idx] = (uint16_t)(split_hi(r1, r2) >> 16);
            o1[P_A1_PART + idx] = (uint16_t)bf16_rne(r2);
        }
        const float wt = W1[fk * 64 + (32 * tp + i)];               // transposed: rows = in feature, K = out feature
        oT[idx] = (uint16_t)(split_hi(wt, r1) >> 16);
        oT[P_A1_PART / 2 + idx] = (uint16_t)bf16_rne(r1);
    }
}

// ===================================================================================================
// forward statistics
// ===================================================================================================
// moments of the kept coordinates over all B*N points: per-workgroup partials (double)
//   part[blk][0..4] = sum xa, sum xb, sum xa^2, sum xb^2, sum xa*xb
__global__ __launch_bounds__(256) void tstats_x_kernel(int N, int ka, int kb, const float *__restrict__ p,
                                                       double *__restrict__ part) {
    __shared__ double red[4][5];
    const int bi = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    double v[5] = {0, 0, 0, 0, 0};
    if (n < N) {
        const float *pc = p + (size_t)bi * 3 * N;
        const double xa = pc[(size_t)ka * N + n], xb = kb >= 0 ? pc[(size_t)kb * N + n] : 0.0;
        v[0] = xa; v[1] = xb; v[2] = xa * xa; v[3] = xb * xb; v[4] = xa * xb;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 5; ++i) red[threadIdx.x >> 6][i] = v[i];
    __syncthreads();
    if (threadIdx.x < 5)
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// BN0 batch statistics (analytic: h0 = W0 x is linear in x) and the two folded input-MFMA fragment
// sets of the layer.  One workgroup, thread = (branch, feature).
__global__ __launch_bounds__(128) void tbn0_kernel(int nblk, double count, const double *__restrict__ part,
                                                   const float *__restrict__ tcanon_l, uint8_t *__restrict__ packed_a0,
                                                   float *__restrict__ stats_l) {
    __shared__ double mom[5];
    __shared__ float fold[2][64][4];     // per (branch, feature): w_a', w_b', T'  |  and normalised variants share the loop
    __shared__ float foldn[2][64][4];
    if (threadIdx.x < 5) {
        double s = 0;
        for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 8 + threadIdx.x];
        mom[threadIdx.x] = s / count;
    }
    __syncthreads();
    const int br = threadIdx.x >> 6, f = threadIdx.x & 63;
    const float *cb = tcanon_l + br * T_BR;
    const double wa = cb[T_W0 + f * 2 + 0], wb = cb[T_W0 + f * 2 + 1];
    const double ea = mom[0], eb = mom[1];
    const double caa = mom[2] - ea * ea, cbb = mom[3] - eb * eb, cab = mom[4] - ea * eb;
    const double mean = wa * ea + wb * eb;
    double var = wa * wa * caa + wb * wb * cbb + 2.0 * wa * wb * cab;     // biased, as BatchNorm normalises with
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const float gamma = cb[T_G0 + f], beta = cb[T_B0 + f];
    float *st = stats_l + br * ST_BR;
    st[0 * 64 + f] = (float)mean;
    st[1 * 64 + f] = rstd;
    st[4 * 64 + f] = (float)(var * (count / (count > 1 ? count - 1 : 1)));      // unbiased: running_var update
    const float s0 = gamma * rstd;
    fold[br][f][0] = s0 * (float)wa; fold[br][f][1] = s0 * (float)wb; fold[br][f][2] = beta - (float)mean * s0;
    foldn[br][f][0] = rstd * (float)wa; foldn[br][f][1] = rstd * (float)wb; foldn[br][f][2] = -(float)mean * rstd;
    __syncthreads();
    uint16_t *a0 = (uint16_t *)packed_a0, *a0n = (uint16_t *)(packed_a0 + 4096);
    for (int idx = threadIdx.x; idx < 2 * 2 * 64 * 8; idx += blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) & 1, b2 = idx >> 10;
        const int ff = 32 * t + (lane & 31), h = lane >> 5;
        a0[idx] = (uint16_t)input_weight_slot(fold[b2][ff][h], fold[b2][ff][2], h, j);
        a0n[idx] = (uint16_t)input_weight_slot(foldn[b2][ff][h], foldn[b2][ff][2], h, j);
    }
}

// ---------------------------------------------------------------------------------------------------
// shared tile machinery (one 32-point tile per wave, weights staged in LDS)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_bytes(const uint8_t *src, uint8_t *lds, int nbytes, int wave, int lane) {
    for (int c = wave; c * 1024 < nbytes; c += TW)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + c * 1024 + lane * 16), (lds_void *)(lds + c * 1024), 16, 0, 0);
}

// input MFMA of one branch: acc[t] = A0[br][t] . b0   (t = M tile)
__device__ __forceinline__ void input_mfma(const uint8_t *a0, int br, int lane, u32x4 b0, f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
        acc[t] = mfma(*(const u32x4 *)(a0 + ((br * 2 + t) * 64 + lane) * 16), b0, zero16());
}

// bf16 split (NS parts) of an accumulator fragment pair into the B fragments of the next contraction
// (register r of M tile t = element j = r&7 of k-step 2t + (r>>3)); RELU = clamp at zero first.
// Same arithmetic as branch_tile in csrc/flow.hip: the recomputed activations are bit-identical to
// the forward kernel's.
template <bool RELU, int NS>
__device__ __forceinline__ void split_fragment(const f32x16 (&v)[2], u32x4 (&bf)[NS][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = RELU ? relu(v[t][r]) : v[t][r], v1 = RELU ? relu(v[t][r + 1]) : v[t][r + 1];
            const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;
            float l0, l1;
            split_hi(v0, l0); split_hi(v1, l1);
            bf[0][s][d] = pack_bf16_trunc(v0, v1);
            if constexpr (NS == 2) {
                bf[1][s][d] = pack_bf16_rne(l0, l1);
            } else {
                float m0, m1;
                split_hi(l0, m0); split_hi(l1, m1);
                bf[1][s][d] = pack_bf16_trunc(l0, l1);
                bf[2][s][d] = pack_bf16_rne(m0, m1);
            }
        }
}

// acc[tp] += A1[br] . B  with the split terms of Terms<NS>; a1 = base of [part][br][tp][s][lane]
template <int NS>
__device__ __forceinline__ void chain_mfma(const uint8_t *a1, int br, int lane, const u32x4 (&bf)[NS][4], f32x16 (&acc)[2]) {
    using TT = Terms<NS>;
#pragma unroll
    for (int term = 0; term < TT::N; ++term)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 af[2];
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                af[tp] = *(const u32x4 *)(a1 + TT::A[term] * P_A1_PART + (((br * 2 + tp) * 4 + s) * 64 + lane) * 16);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) acc[tp] = mfma(af[tp], bf[TT::B[term]][s], acc[tp]);
        }
}

// per-lane vector of a per-feature LDS array for the features this lane holds: out[t][r]
__device__ __forceinline__ void load_features(const float *vec, int h, f32x16 (&out)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4 *)(vec + 32 * t + 8 * q + 4 * h);
            out[t][4 * q + 0] = v.x; out[t][4 * q + 1] = v.y; out[t][4 * q + 2] = v.z; out[t][4 * q + 3] = v.w;
        }
}

// Sum over the 32 point-lanes of a half wave, for all 32 accumulator registers at once: a
// recursive-halving butterfly (31 shuffles instead of 160).  On return lane pl holds in v[0][0] the
// total of register index R(pl) = b0*16 + b1*8 + b2*4 + b3*2 + b4 (b_k = bit k of pl), i.e. of
// feature acc_feature(R >> 4, R & 15, h).
template <int K>
__device__ __forceinline__ void reduce_stage(float (&w)[32], int pl) {
    constexpr int n2 = 16 >> K;
    const bool up = (pl >> K) & 1;
#pragma unroll
    for (int i = 0; i < n2; ++i) {
        const float keep = up ? w[i + n2] : w[i];
        const float send = up ? w[i] : w[i + n2];
        w[i] = keep + __shfl_xor(send, 1 << K);
    }
}
__device__ __forceinline__ float reduce_points(const f32x16 (&v)[2], int pl) {
    float w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = v[i >> 4][i & 15];
    reduce_stage<0>(w, pl); reduce_stage<1>(w, pl); reduce_stage<2>(w, pl); reduce_stage<3>(w, pl); reduce_stage<4>(w, pl);
    return w[0];
}
__device__ __forceinline__ int reduced_feature(int pl, int h) {
    const int R = ((pl & 1) << 4) | ((pl & 2) << 2) | (pl & 4) | ((pl & 8) >> 2) | ((pl & 16) >> 4);
    return acc_feature(R >> 4, R & 15, h);
}

// raw dump / reload of an accumulator fragment pair (gradients that cross a grid-wide reduction)
__device__ __forceinline__ void dump_fragment(float *buf, size_t tile, int br, int lane, const f32x16 (&v)[2]) {
    float *o = buf + ((tile * 2 + br) * 32) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i * 64] = v[i >> 4][i & 15];
}
__device__ __forceinline__ void load_fragment(const float *buf, size_t tile, int br, int lane, f32x16 (&v)[2]) {
    const float *o = buf + ((tile * 2 + br) * 32) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i >> 4][i & 15] = o[i * 64];
}

struct TArgs {
    const uint8_t *packed_l;     // PT_BYTES of this layer
    const float *tcanon_l;       // T_LAYER floats
    const float *film_l;         // (B, 512) eval FiLM blocks of this layer
    const float *filmb_l;        // (B, FB_CLOUD) backward FiLM blocks
    const float *stats_l;        // ST_LAYER
    const float *p_in;           // (B, 3, N)
    int B, N, ka, kb, wa, wb, mode, a0_off;
    float eps;
};

// h1 = W1 relu(BN0(W0 x)) for every point; per-workgroup partial sums and sums of squares
//   part[blk][br][2][64]
template <int NS>
__global__ __launch_bounds__(TW * 64) void tstats_h1_kernel(TArgs a, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ float acc_s[2][2][64];
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_bytes(a.packed_l, smem, pt_a0n(NS), wave, lane);                // A1 + A0
    if (threadIdx.x < 256) ((float *)acc_s)[threadIdx.x] = 0.f;
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const float *pc = a.p_in + (size_t)bi * 3 * N;
    const int nc = valid ? n : N - 1;
    const float xa = pc[(size_t)a.ka * N + nc], xb = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    __syncthreads();
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        f32x16 acc0[2], acc1[2] = {zero16(), zero16()}, sq[2];
        u32x4 bf[NS][4];
        input_mfma(smem + pt_a0(NS), br, lane, b0, acc0);
        split_fragment<true, NS>(acc0, bf);
        chain_mfma<NS>(smem + PT_A1, br, lane, bf, acc1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc1[t][r] = valid ? acc1[t][r] : 0.f;                   // padding lanes do not count
                sq[t][r] = acc1[t][r] * acc1[t][r];
            }
        const float s1 = reduce_points(acc1, pl), s2 = reduce_points(sq, pl);
        const int f = reduced_feature(pl, h);
        atomicAdd(&acc_s[br][0][f], s1);                                  // LDS atomics: 8 waves per address
        atomicAdd(&acc_s[br][1][f], s2);
    }
    __syncthreads();
    if (threadIdx.x < 256) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = ((float *)acc_s)[threadIdx.x];
}

// BN1 batch statistics from the partial sums, then this step's FiLM fold for one cloud:
//   eval block  (csrc/flow.hip film layout): D = FC/FA, W2' = W2*FA, b2
//   bwd block   a = eps + e^cw, c = cb, rstd1, c/a
// fm_l: [br][sub(w,b)][B][64]
__global__ __launch_bounds__(128) void tfilm_fold_kernel(int nblk, float count, const float *__restrict__ part,
                                                         const float *__restrict__ tcanon_l, const float *__restrict__ fm_l,
                                                         int B, float eps, float *__restrict__ stats_l,
                                                         float *__restrict__ film_l, float *__restrict__ filmb_l) {
    const int b = blockIdx.x, br = threadIdx.x >> 6, f = threadIdx.x & 63;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < nblk; ++k) {
        s1 += part[(size_t)k * 256 + (br * 2 + 0) * 64 + f];
        s2 += part[(size_t)k * 256 + (br * 2 + 1) * 64 + f];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (b == 0) {
        float *st = stats_l + br * ST_BR;
        st[2 * 64 + f] = (float)mean;
        st[3 * 64 + f] = rstd;
        st[5 * 64 + f] = (float)(var * (count / (count > 1 ? count - 1 : 1)));
    }
    const float cw = fm_l[((size_t)(br * 2 + 0) * B + b) * 64 + f], cb = fm_l[((size_t)(br * 2 + 1) * B + b) * 64 + f];
    const float av = eps + expf(cw);
    const float FA = av * rstd, FC = -av * (float)mean * rstd + cb;
    const float *cbp = tcanon_l + br * T_BR;
    float *o = film_l + (size_t)b * (FILM_BYTES / 4) + br * FILM_BR_FLOATS;
    o[f] = FC / FA;
    o[64 + 2 * f] = cbp[T_W2 + f] * FA;
    o[64 + 2 * f + 1] = cbp[T_W2 + 64 + f] * FA;
    if (f < 2) film_l[(size_t)b * (FILM_BYTES / 4) + FILM_B2_OFF + br * 2 + f] = cbp[T_B2 + f];
    float *ob = filmb_l + (size_t)b * FB_CLOUD + br * FB_BR;
    ob[0 * 64 + f] = av;
    ob[1 * 64 + f] = cb;
    ob[2 * 64 + f] = rstd;
    ob[3 * 64 + f] = cb / av;
}

// ===================================================================================================
// backward
// ===================================================================================================
// LDS map of the backward kernels (bytes)
constexpr int L_PACK = 0;                                                  // packed layer block (pt_bytes)
__host__ __device__ constexpr int l_film(int NS) { return pt_bytes(NS); }            // eval FiLM block of this cloud (2048)
__host__ __device__ constexpr int l_filmb(int NS) { return l_film(NS) + 2048; }      // backward FiLM block (2048)
__host__ __device__ constexpr int l_red(int NS) { return l_filmb(NS) + 2048; }       // workgroup reduction scratch
constexpr int XY_WAVE = 2 * 64 * 32;                                       // bf16 elements of one wave's X | Y tiles

// Pass 1: recompute the layer to h2, differentiate the coupling transform and the output SharedDot.
//   stores  dh2a fragments (scratch)         dp_in <- direct term  g * d(p_out)/d(p)
//   partial sums per workgroup: part1[blk][br][k][64], k = 0 dW2a, 1 dW2b, 2 da, 3 dc;  part1b[blk][br][2] = db2
template <int NS>
__global__ __launch_bounds__(TW * 64) void tbwd1_kernel(TArgs a, const float *__restrict__ g_p, const float *__restrict__ g_mu,
                                                        const float *__restrict__ g_lv, float *__restrict__ dp_in,
                                                        float *__restrict__ scratch, float *__restrict__ part1,
                                                        float *__restrict__ part1b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int L_FILM = l_film(NS), L_FILMB = l_filmb(NS), L_RED = l_red(NS);
    float *red = (float *)(smem + L_RED);                                  // [2 br][4][64] + [2][2]
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_bytes(a.packed_l, smem + L_PACK, pt_a0n(NS), wave, lane);
    stage_bytes((const uint8_t *)(a.film_l + (size_t)bi * 512), smem + L_FILM, 2048, wave, lane);
    stage_bytes((const uint8_t *)(a.filmb_l + (size_t)bi * FB_CLOUD), smem + L_FILMB, 2048, wave, lane);
    for (int i = threadIdx.x; i < 2 * 4 * 64 + 4; i += TW * 64) red[i] = 0.f;
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const size_t cloud = (size_t)bi * 3 * N;
    float p[3], gp[3], gm[3], gl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p[c] = a.p_in[cloud + (size_t)c * N + nc];
        gp[c] = valid ? g_p[cloud + (size_t)c * N + nc] : 0.f;
        gm[c] = valid && g_mu ? g_mu[cloud + (size_t)c * N + nc] : 0.f;
        gl[c] = valid && g_lv ? g_lv[cloud + (size_t)c * N + nc] : 0.f;
    }
    const float xa = sel3(a.ka, p[0], p[1], p[2]), xb = a.kb >= 0 ? sel3(a.kb, p[0], p[1], p[2]) : 0.f;
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    __syncthreads();
    const float *film = (const float *)(smem + L_FILM);
    const float *filmb = (const float *)(smem + L_FILMB);
    // ---- forward recompute of both branches: o[br][w], and the h2 pre-activations
    f32x16 pre[2][2];                                                      // acc1 = h1 + D  (h2a = FA * acc1)
    float o[2][2];
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        f32x16 acc0[2];
        u32x4 bf[NS][4];
        input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, acc0);
        split_fragment<true, NS>(acc0, bf);
        load_features(film + br * FILM_BR_FLOATS, h, pre[br]);             // accumulator starts at D
        chain_mfma<NS>(smem + L_PACK + PT_A1, br, lane, bf, pre[br]);
        float oa = 0.f, ob = 0.f;
        const float *wab = film + br * FILM_BR_FLOATS + 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = acc_feature(t, r, 0) + 4 * h;
                const float v = relu(pre[br][t][r]);
                oa += wab[2 * f] * v; ob += wab[2 * f + 1] * v;
            }
        oa += __shfl_xor(oa, 32); ob += __shfl_xor(ob, 32);
        o[br][0] = oa + film[FILM_B2_OFF + br * 2 + 0];
        o[br][1] = ob + film[FILM_B2_OFF + br * 2 + 1];
    }
    // ---- coupling transform and its derivative (flows.py:96-115)
    const bool inverse = a.mode == DPF_MODE_INVERSE;
    float dmu_w[2] = {0.f, 0.f}, dlv_w[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool isa = c == a.wa, isb = c == a.wb;
        const float olv = isa ? o[0][0] : (isb ? o[0][1] : 0.f);
        const float lv = olv / (1.0f + fabsf(olv));
        const float mu = isa ? o[1][0] : (isb ? o[1][1] : 0.f);
        const float e = expf(lv), var = a.eps + e;
        float dmu, dlv, dpc;
        if (inverse) {
            const float r = 1.0f / sqrtf(var);
            dpc = gp[c] * r;
            dmu = gm[c] - gp[c] * r;
            dlv = gl[c] + gp[c] * (p[c] - mu) * (-0.5f * e * r * r * r);
        } else {
            const float s = sqrtf(var);
            dpc = gp[c] * s;
            dmu = gm[c] + gp[c];
            dlv = gl[c] + gp[c] * p[c] * (0.5f * e / s);
        }
        if (valid && h == 0) dp_in[cloud + (size_t)c * N + n] = dpc;       // direct term; pass 3 adds the conditioner path
        const float dsoft = 1.0f / ((1.0f + fabsf(olv)) * (1.0f + fabsf(olv)));
        if (isa) { dmu_w[0] = dmu; dlv_w[0] = dlv * dsoft; }
        if (isb) { dmu_w[1] = dmu; dlv_w[1] = dlv * dsoft; }
    }
    // ---- output SharedDot backward, FiLM backward, per-feature sums
    const size_t tile = ((size_t)bi * gridDim.x + blockIdx.x) * TW + wave;
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const float doa = br == 0 ? dlv_w[0] : dmu_w[0], dob = br == 0 ? dlv_w[1] : dmu_w[1];
        const float *wab = film + br * FILM_BR_FLOATS + 64;
        const float *fb = filmb + br * FB_BR;
        f32x16 dh2a[2], tW2a[2], tW2b[2], tda[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = acc_feature(t, r, 0) + 4 * h;
                const float pa = pre[br][t][r];
                const bool on = pa > 0.f;
                // h2 = FA*relu(pa); W2' = W2*FA  ->  dh2a = sum_w W2[w]*do_w * [on] = (W2'[w]/FA)*do_w ; dW2[w] = do_w*h2
                const float FA = fb[0 * 64 + f] * fb[2 * 64 + f];                  // a * rstd1
                const float g2 = on ? (wab[2 * f] * doa + wab[2 * f + 1] * dob) / FA : 0.f;    // d/d(h2a)
                dh2a[t][r] = g2;
                const float h2 = on ? FA * pa : 0.f;
                tW2a[t][r] = doa * h2;
                tW2b[t][r] = dob * h2;
                // h1n = pa*rstd1 - c/a
                tda[t][r] = g2 * (pa * fb[2 * 64 + f] - fb[3 * 64 + f]);
            }
        dump_fragment(scratch, tile, br, lane, dh2a);
        const int f = reduced_feature(pl, h);
        const float r0 = reduce_points(tW2a, pl), r1 = reduce_points(tW2b, pl), r2 = reduce_points(tda, pl),
                    r3 = reduce_points(dh2a, pl);
        atomicAdd(&red[(br * 4 + 0) * 64 + f], r0);
        atomicAdd(&red[(br * 4 + 1) * 64 + f], r1);
        atomicAdd(&red[(br * 4 + 2) * 64 + f], r2);
        atomicAdd(&red[(br * 4 + 3) * 64 + f], r3);
        float sa = h == 0 ? doa : 0.f, sb = h == 0 ? dob : 0.f;                   // db2: each point once
        for (int q = 32; q > 0; q >>= 1) { sa += __shfl_xor(sa, q); sb += __shfl_xor(sb, q); }
        if (lane == 0) { atomicAdd(&red[512 + br * 2 + 0], sa); atomicAdd(&red[512 + br * 2 + 1], sb); }
    }
    __syncthreads();
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 512; i += TW * 64) part1[blk * 512 + i] = red[i];
    if (threadIdx.x < 4) part1b[blk * 4 + threadIdx.x] = red[512 + threadIdx.x];
}

// reduce per-workgroup partials: out[j] = scale * sum_k part[k*J + j]   (deterministic)
__global__ __launch_bounds__(256) void treduce_kernel(int nblk, int J, const float *__restrict__ part, float *__restrict__ out,
                                                      float scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    double s = 0;
    for (int k = 0; k < nblk; ++k) s += part[(size_t)k * J + j];
    out[j] = (float)(s * scale);
}

// per-cloud reduce: out[b][j] = sum over the nb workgroups of cloud b
__global__ __launch_bounds__(256) void treduce_cloud_kernel(int nb, int J, const float *__restrict__ part, float *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= J) return;
    float s = 0;
    for (int k = 0; k < nb; ++k) s += part[((size_t)b * nb + k) * J + j];
    out[(size_t)b * J + j] = s;
}

// Finish pass 1 on the host side of the reductions: the FiLM gradients and the BN1 sums.
//   part1 totals per cloud pc[b][br][k][64] (k: 0 dW2a 1 dW2b 2 da 3 dc)
//   -> dfm_l[br][sub][b][64] = (da * e^cw, dc);  s12[br][2][64] = (sum dh1n, sum dh1n*h1n) / P
//   -> dW2, db2 into dcanon_l
__global__ __launch_bounds__(128) void tbwd1_finish_kernel(int B, int nb, float count, const float *__restrict__ pc,
                                                           const float *__restrict__ part1b, int nblk,
                                                           const float *__restrict__ filmb_l, float eps,
                                                           float *__restrict__ dfm_l, float *__restrict__ s12,
                                                           float *__restrict__ dcanon_l) {
    const int br = threadIdx.x >> 6, f = threadIdx.x & 63;
    double S1 = 0, S2 = 0, w2a = 0, w2b = 0;
    for (int b = 0; b < B; ++b) {
        const float *q = pc + ((size_t)b * 2 + br) * 256;
        const float da = q[2 * 64 + f], dc = q[3 * 64 + f];
        const float av = filmb_l[(size_t)b * FB_CLOUD + br * FB_BR + f];
        dfm_l[((size_t)(br * 2 + 0) * B + b) * 64 + f] = da * (av - eps);       // d cw = da * e^cw
        dfm_l[((size_t)(br * 2 + 1) * B + b) * 64 + f] = dc;
        S1 += (double)av * dc;                                                   // dh1n = a * dh2a
        S2 += (double)av * da;                                                   // dh1n * h1n
        w2a += q[0 * 64 + f]; w2b += q[1 * 64 + f];
    }
    s12[(br * 2 + 0) * 64 + f] = (float)(S1 / count);
    s12[(br * 2 + 1) * 64 + f] = (float)(S2 / count);
    dcanon_l[br * T_BR + T_W2 + f] = (float)w2a;
    dcanon_l[br * T_BR + T_W2 + 64 + f] = (float)w2b;
    if (f < 4) {
        double s = 0;
        if (f < 2) for (int k = 0; k < nblk; ++k) s += part1b[(size_t)k * 4 + br * 2 + f];
        dcanon_l[br * T_BR + T_B2 + f] = (float)s;
    }
}

// Pass 2: BN1 backward, dh0 = W1^T dh1 (matrix cores), dW1 = dh1 h0^T (matrix cores, contraction over the
// tile's 32 points through an LDS transpose), relu backward; stores dh0a; partials of d gamma0 / d beta0.
//   part2[blk][br][4224]: [0..63] d gamma0, [64..127] d beta0, [128..4223] dW1 (row = out feature)
template <int NS>
__global__ __launch_bounds__(TW * 64) void tbwd2_kernel(TArgs a, const float *__restrict__ s12, float *__restrict__ scratch,
                                                        float *__restrict__ scratch2, float *__restrict__ part2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int L_FILM = l_film(NS), L_FILMB = l_filmb(NS), L_RED = l_red(NS);
    float *red = (float *)(smem + L_RED);                                  // [128] d gamma0 | d beta0 of the branch
    uint16_t *tr = (uint16_t *)(smem + L_RED + 512);                       // per wave: X[64][32], Y[64][32] bf16, swizzled
    float *redw = (float *)(smem + L_RED + 512);                           // [4096] dW1, ALIASES tr once the MFMAs are done
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_bytes(a.packed_l, smem + L_PACK, pt_bytes(NS), wave, lane);
    stage_bytes((const uint8_t *)(a.film_l + (size_t)bi * 512), smem + L_FILM, 2048, wave, lane);
    stage_bytes((const uint8_t *)(a.filmb_l + (size_t)bi * FB_CLOUD), smem + L_FILMB, 2048, wave, lane);
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const float *pc = a.p_in + (size_t)bi * 3 * N;
    const float xa = pc[(size_t)a.ka * N + nc], xb = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    const float *film = (const float *)(smem + L_FILM);
    const float *filmb = (const float *)(smem + L_FILMB);
    const size_t tile = ((size_t)bi * gridDim.x + blockIdx.x) * TW + wave;
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    uint16_t *X = tr + wave * XY_WAVE, *Y = X + 64 * 32;
    // tile element (row f, point q) lives at f*32 + ((q>>3) ^ ((f>>2)&3))*8 + (q&7): the 16-byte chunks of a
    // row are XOR-swizzled so that the 16 rows a ds_read_b128 group touches hit 16 distinct bank quads
    auto xy_off = [](int f, int chunk) { return f * 32 + ((chunk ^ ((f >> 2) & 3)) << 3); };
    for (int br = 0; br < 2; ++br) {
        __syncthreads();                                                   // staging landed / previous branch flushed
        if (threadIdx.x < 128) red[threadIdx.x] = 0.f;
        __syncthreads();
        f32x16 h0a[2], h0n[2], pre[2], g2[2];
        u32x4 bf[NS][4];
        input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, h0a);          // gamma*h0n + beta
        input_mfma(smem + L_PACK + pt_a0n(NS), br, lane, b0, h0n);         // normalised
        split_fragment<true, NS>(h0a, bf);
        load_features(film + br * FILM_BR_FLOATS, h, pre);
        chain_mfma<NS>(smem + L_PACK + PT_A1, br, lane, bf, pre);          // pre = h1 + D
        load_fragment(scratch, tile, br, lane, g2);                        // dh2a from pass 1
        const float *fb = filmb + br * FB_BR;
        // dh1 = rstd1 * (dh1n - mean(dh1n) - h1n * mean(dh1n*h1n)),  dh1n = a*dh2a,  h1n = pre*rstd1 - c/a
        f32x16 dh1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = acc_feature(t, r, 0) + 4 * h;
                const float rstd1 = fb[2 * 64 + f];
                const float h1n = pre[t][r] * rstd1 - fb[3 * 64 + f];
                const float dh1n = fb[0 * 64 + f] * g2[t][r];
                const float v = rstd1 * (dh1n - s12[(br * 2 + 0) * 64 + f] - h1n * s12[(br * 2 + 1) * 64 + f]);
                dh1[t][r] = valid ? v : 0.f;
            }
        // ---- dh0 = W1^T dh1
        u32x4 bg[2][4];
        f32x16 dh0[2] = {zero16(), zero16()};
        split_fragment<false, 2>(dh1, bg);
        chain_mfma<2>(smem + L_PACK + pt_a1t(NS), br, lane, bg, dh0);
        f32x16 dh0a[2], tg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = h0a[t][r] > 0.f ? dh0[t][r] : 0.f;
                dh0a[t][r] = v;
                tg[t][r] = v * h0n[t][r];
            }
        dump_fragment(scratch2, tile, br, lane, dh0a);
        {
            const int f = reduced_feature(pl, h);
            const float r0 = reduce_points(tg, pl), r1 = reduce_points(dh0a, pl);
            atomicAdd(&red[f], r0);
            atomicAdd(&red[64 + f], r1);
        }
        // ---- dW1[fo][fi] += sum_points dh1[fo][pt] * h0[fi][pt]: both fragments are transposed through LDS so
        // that the tile's 32 points become the K dimension (2 k-steps of 16).  hi/lo split like every other
        // contraction: three rounds (hi.hi, hi.lo, lo.hi) over the same two LDS tiles.
        f32x16 dw[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};     // [fo tile][fi tile]
        auto put = [&](uint16_t *dst, const f32x16 (&v)[2], bool clamp, bool lo) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = acc_feature(t, r, 0) + 4 * h;
                    const float x = clamp ? relu(v[t][r]) : v[t][r];
                    float rest;
                    const uint32_t hi = split_hi(x, rest);
                    dst[xy_off(f, pl >> 3) + (pl & 7)] = lo ? (uint16_t)bf16_rne(rest) : (uint16_t)(hi >> 16);
                }
        };
        auto outer = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fa[2], fbb[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    fa[mt] = *(const u32x4 *)(X + xy_off(32 * mt + pl, 2 * ks + h));     // row fo, 8 consecutive points
                    fbb[mt] = *(const u32x4 *)(Y + xy_off(32 * mt + pl, 2 * ks + h));    // col fi, same points
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) dw[mt][nt] = mfma(fa[mt], fbb[nt], dw[mt][nt]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        put(X, dh1, false, false); put(Y, h0a, true, false); outer();       // hi . hi
        put(Y, h0a, true, true); outer();                                   // hi . lo
        put(X, dh1, false, true); put(Y, h0a, true, false); outer();        // lo . hi
        __syncthreads();                                                   // every wave is done with its X / Y
        for (int i = threadIdx.x; i < 4096; i += TW * 64) redw[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fo = acc_feature(mt, r, 0) + 4 * h, fi = 32 * nt + pl;
                    atomicAdd(&redw[fo * 64 + fi], dw[mt][nt][r]);
                }
        __syncthreads();
        float *o = part2 + (blk * 2 + br) * 4224;
        if (threadIdx.x < 128) o[threadIdx.x] = red[threadIdx.x];
        for (int i = threadIdx.x; i < 4096; i += TW * 64) o[128 + i] = redw[i];
    }
}

// Pass 3: BN0 backward (batch statistics), dW0, and the conditioner path of d(input points).
//   tot[br][0..63] = sum dh0a*h0n (d gamma0), [64..127] = sum dh0a (d beta0)   (already reduced, from dcanon_l)
//   part3[blk][br][128]: dW0 [64][2]
__global__ __launch_bounds__(TW * 64) void tbwd3_kernel(TArgs a, float count, const float *__restrict__ dcanon_l,
                                                        const float *__restrict__ scratch2, float *__restrict__ dp_in,
                                                        float *__restrict__ part3) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *red = (float *)(smem + 8192);                                   // [2 br][128]
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_bytes(a.packed_l + a.a0_off, smem, 8192, wave, lane);            // A0 and A0N only
    for (int i = threadIdx.x; i < 256; i += TW * 64) red[i] = 0.f;
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const float *pc = a.p_in + (size_t)bi * 3 * N;
    const float xa = pc[(size_t)a.ka * N + nc], xb = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    const size_t tile = ((size_t)bi * gridDim.x + blockIdx.x) * TW + wave;
    __syncthreads();
    float dxa = 0.f, dxb = 0.f;
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        f32x16 h0n[2], g[2], ta[2], tb[2];
        input_mfma(smem + 4096, br, lane, b0, h0n);
        load_fragment(scratch2, tile, br, lane, g);                        // dh0a from pass 2
        const float *cb = a.tcanon_l + br * T_BR;
        const float *st = a.stats_l + br * ST_BR;
        const float *dg = dcanon_l + br * T_BR;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = acc_feature(t, r, 0) + 4 * h;
                const float gamma = cb[T_G0 + f], rstd0 = st[1 * 64 + f];
                // dh0n = gamma*dh0a; mean(dh0n) = gamma*dbeta/P; mean(dh0n*h0n) = gamma*dgamma/P
                const float v = rstd0 * gamma * (g[t][r] - dg[T_B0 + f] / count - h0n[t][r] * dg[T_G0 + f] / count);
                const float dpre = valid ? v : 0.f;                        // d h0pre
                ta[t][r] = dpre * xa;
                tb[t][r] = dpre * xb;
                dxa += cb[T_W0 + f * 2 + 0] * dpre;
                dxb += cb[T_W0 + f * 2 + 1] * dpre;
            }
        const int f = reduced_feature(pl, h);
        const float r0 = reduce_points(ta, pl), r1 = reduce_points(tb, pl);
        atomicAdd(&red[br * 128 + f * 2 + 0], r0);
        atomicAdd(&red[br * 128 + f * 2 + 1], r1);
    }
    dxa += __shfl_xor(dxa, 32); dxb += __shfl_xor(dxb, 32);
    if (valid && h == 0) {
        float *d = dp_in + (size_t)bi * 3 * N;
        d[(size_t)a.ka * N + n] += dxa;
        if (a.kb >= 0) d[(size_t)a.kb * N + n] += dxb;
    }
    __syncthreads();
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 256; i += TW * 64) part3[blk * 256 + i] = red[i];
}

// strided copy helper for the reduce outputs: part2 totals [br][4224] -> dcanon (gamma0, beta0, W1 are contiguous
// at T_G0 .. T_W2), part3 totals [br][128] -> dcanon T_W0
__global__ __launch_bounds__(256) void treduce_to_canon_kernel(int nblk, int J, int dst_off, const float *__restrict__ part,
                                                               float *__restrict__ dcanon_l) {
    const int j = blockIdx.x * 256 + threadIdx.x, br = blockIdx.y;
    if (j >= J) return;
    double s = 0;
    for (int k = 0; k < nblk; ++k) s += part[((size_t)k * 2 + br) * J + j];
    dcanon_l[br * T_BR + dst_off + j] = (float)s;
}

hipError_t set_lds(const void *fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

static inline int t_ns(int precision) { return precision == DPF_PREC_BF16X3 ? 2 : (precision == DPF_PREC_BF16X6 ? 3 : 0); }

extern "C" size_t dpf_flow_train_canon_floats(void) { return (size_t)T_LAYER; }
extern "C" size_t dpf_flow_train_packed_bytes(int n_layers, int precision) {
    return t_ns(precision) ? (size_t)n_layers * pt_bytes(t_ns(precision)) : 0;
}
extern "C" size_t dpf_flow_train_stats_floats(void) { return (size_t)ST_LAYER; }
extern "C" size_t dpf_flow_train_film_floats(int B) { return (size_t)B * (512 + FB_CLOUD); }

static inline int t_nblk(int B, int N) { return B * ((N + TBLK - 1) / TBLK); }

// workspace (floats) for one layer call, forward or backward
extern "C" size_t dpf_flow_train_workspace_bytes(int B, int N) {
    const size_t nblk = (size_t)t_nblk(B, N), nbx = (size_t)B * ((N + 255) / 256);
    size_t f = 0;
    f += nbx * 8 * 2;                 // x-moment partials (double)
    f += nblk * 512;                  // h1 / pass-1 partials
    f += nblk * 4;                    // db2 partials
    f += (size_t)B * 512;             // per-cloud totals of pass 1
    f += 256;                         // s12
    f += nblk * 2 * 4224;             // pass-2 partials
    f += nblk * 256;                  // pass-3 partials
    return f * sizeof(float) + 1024;
}
// the two fragment scratch buffers (dh2a, dh0a): floats
extern "C" size_t dpf_flow_train_scratch_floats(int B, int N) { return (size_t)t_nblk(B, N) * TW * 2 * 32 * 64; }

extern "C" int dpf_flow_train_pack(int n_layers, int precision, const float *tcanon, void *packed, dpf_stream_t stream) {
    const int ns = t_ns(precision);
    if (n_layers <= 0 || !tcanon || !packed) return DPF_EINVAL;
    if (!ns) return DPF_ENOSUP;
    if (ns == 2) hipLaunchKernelGGL(tpack_kernel<2>, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, tcanon, (uint8_t *)packed);
    else hipLaunchKernelGGL(tpack_kernel<3>, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, tcanon, (uint8_t *)packed);
    return (int)hipGetLastError();
}

struct TWork {
    double *xpart; float *part1, *part1b, *pc, *s12, *part2, *part3;
};
static TWork carve(void *ws, int B, int N) {
    const size_t nblk = (size_t)t_nblk(B, N), nbx = (size_t)B * ((N + 255) / 256);
    TWork w;
    uint8_t *p = (uint8_t *)ws;
    w.xpart = (double *)p; p += nbx * 8 * sizeof(double);
    w.part1 = (float *)p; p += nblk * 512 * 4;
    w.part1b = (float *)p; p += nblk * 4 * 4;
    w.pc = (float *)p; p += (size_t)B * 512 * 4;
    w.s12 = (float *)p; p += 256 * 4;
    w.part2 = (float *)p; p += nblk * 2 * 4224 * 4;
    w.part3 = (float *)p;
    return w;
}

template <int NS>
static int prepare_layer(int B, int N, int ka, int kb, const float *tcanon_l, void *packed_l, const float *fm_l,
                         const float *p_in, float *stats_l, float *film_l, float flow_eps, void *workspace, hipStream_t s) {
    TWork w = carve(workspace, B, N);
    const int nbx = (N + 255) / 256;
    const double count = (double)B * N;
    hipLaunchKernelGGL(tstats_x_kernel, dim3(nbx, B), dim3(256), 0, s, N, ka, kb, p_in, w.xpart);
    hipLaunchKernelGGL(tbn0_kernel, dim3(1), dim3(128), 0, s, nbx * B, count, w.xpart, tcanon_l,
                       (uint8_t *)packed_l + pt_a0(NS), stats_l);
    TArgs a;
    a.packed_l = (const uint8_t *)packed_l; a.tcanon_l = tcanon_l; a.film_l = film_l; a.filmb_l = film_l + (size_t)B * 512;
    a.stats_l = stats_l; a.p_in = p_in; a.B = B; a.N = N; a.ka = ka; a.kb = kb; a.wa = 0; a.wb = 0; a.mode = 0;
    a.a0_off = pt_a0(NS); a.eps = flow_eps;
    static bool attr = false;
    if (!attr) {
        hipError_t e = set_lds((const void *)tstats_h1_kernel<NS>, pt_a0n(NS));
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const dim3 grid((N + TBLK - 1) / TBLK, B);
    hipLaunchKernelGGL(tstats_h1_kernel<NS>, grid, dim3(TW * 64), pt_a0n(NS), s, a, w.part1);
    hipLaunchKernelGGL(tfilm_fold_kernel, dim3(B), dim3(128), 0, s, (int)(grid.x * grid.y), (float)count, w.part1, tcanon_l,
                       fm_l, B, flow_eps, stats_l, film_l, film_l + (size_t)B * 512);
    return (int)hipGetLastError();
}

// Forward statistics + folds of ONE layer; afterwards dpf_flow_forward(n_layers = 1, precision, packed = packed_l,
// film = film_l) runs the layer itself.  stats_l receives the batch statistics (for the running-stat update).
extern "C" int dpf_flow_train_prepare_layer(int B, int N, int precision, int ka, int kb, const float *tcanon_l, void *packed_l,
                                            const float *fm_l, const float *p_in, float *stats_l, float *film_l,
                                            float flow_eps, void *workspace, dpf_stream_t stream) {
    if (B <= 0 || N <= 0 || !tcanon_l || !packed_l || !fm_l || !p_in || !stats_l || !film_l || !workspace) return DPF_EINVAL;
    if (B > 65535) return DPF_ENOSUP;
    switch (t_ns(precision)) {
        case 2: return prepare_layer<2>(B, N, ka, kb, tcanon_l, packed_l, fm_l, p_in, stats_l, film_l, flow_eps, workspace, (hipStream_t)stream);
        case 3: return prepare_layer<3>(B, N, ka, kb, tcanon_l, packed_l, fm_l, p_in, stats_l, film_l, flow_eps, workspace, (hipStream_t)stream);
        default: return DPF_ENOSUP;
    }
}

template <int NS>
static int backward_layer(int B, int N, int mode, int ka, int kb, int wa, int wb, const float *tcanon_l, const void *packed_l,
                          const float *film_l, const float *stats_l, const float *p_in, const float *g_p, const float *g_mu,
                          const float *g_lv, float *dp_in, float *dcanon_l, float *dfm_l, float *scratch_a, float *scratch_b,
                          float flow_eps, void *workspace, hipStream_t s) {
    TWork w = carve(workspace, B, N);
    TArgs a;
    a.packed_l = (const uint8_t *)packed_l; a.tcanon_l = tcanon_l; a.film_l = film_l; a.filmb_l = film_l + (size_t)B * 512;
    a.stats_l = stats_l; a.p_in = p_in; a.B = B; a.N = N; a.ka = ka; a.kb = kb; a.wa = wa; a.wb = wb; a.mode = mode;
    a.a0_off = pt_a0(NS); a.eps = flow_eps;
    const dim3 grid((N + TBLK - 1) / TBLK, B);
    const int nblk = grid.x * grid.y, nb = grid.x;
    const float count = (float)((double)B * N);
    const int lds1 = l_red(NS) + (512 + 4) * 4, lds2 = l_red(NS) + 512 + TW * XY_WAVE * 2, lds3 = 8192 + 256 * 4;
    static bool attr = false;
    if (!attr) {
        hipError_t e = set_lds((const void *)tbwd1_kernel<NS>, lds1);
        if (e == hipSuccess) e = set_lds((const void *)tbwd2_kernel<NS>, lds2);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(tbwd1_kernel<NS>, grid, dim3(TW * 64), lds1, s, a, g_p, g_mu, g_lv, dp_in, scratch_a, w.part1, w.part1b);
    hipLaunchKernelGGL(treduce_cloud_kernel, dim3(2, B), dim3(256), 0, s, nb, 512, w.part1, w.pc);
    hipLaunchKernelGGL(tbwd1_finish_kernel, dim3(1), dim3(128), 0, s, B, nb, count, w.pc, w.part1b, nblk, a.filmb_l, flow_eps,
                       dfm_l, w.s12, dcanon_l);
    hipLaunchKernelGGL(tbwd2_kernel<NS>, grid, dim3(TW * 64), lds2, s, a, w.s12, scratch_a, scratch_b, w.part2);
    hipLaunchKernelGGL(treduce_to_canon_kernel, dim3((4224 + 255) / 256, 2), dim3(256), 0, s, nblk, 4224, T_G0, w.part2, dcanon_l);
    hipLaunchKernelGGL(tbwd3_kernel, grid, dim3(TW * 64), lds3, s, a, count, dcanon_l, scratch_b, dp_in, w.part3);
    hipLaunchKernelGGL(treduce_to_canon_kernel, dim3(1, 2), dim3(256), 0, s, nblk, 128, T_W0, w.part3, dcanon_l);
    return (int)hipGetLastError();
}

// Backward of ONE layer.  g_p / g_mu / g_lv: gradients w.r.t. the layer's outputs (g_mu, g_lv may be NULL);
// dp_in (B,3,N), dcanon_l (T_LAYER) and dfm_l ([br][sub][B][64]) are fully overwritten.
extern "C" int dpf_flow_train_backward_layer(int B, int N, int mode, int precision, int ka, int kb, int wa, int wb,
                                             const float *tcanon_l, const void *packed_l, const float *film_l,
                                             const float *stats_l, const float *p_in, const float *g_p, const float *g_mu,
                                             const float *g_lv, float *dp_in, float *dcanon_l, float *dfm_l,
                                             float *scratch_a, float *scratch_b, float flow_eps, void *workspace,
                                             dpf_stream_t stream) {
    if (B <= 0 || N <= 0 || !tcanon_l || !packed_l || !film_l || !stats_l || !p_in || !g_p || !dp_in || !dcanon_l || !dfm_l ||
        !scratch_a || !scratch_b || !workspace)
        return DPF_EINVAL;
    if (B > 65535) return DPF_ENOSUP;
#define DPF_BWD(NSV)                                                                                                       \
    return backward_layer<NSV>(B, N, mode, ka, kb, wa, wb, tcanon_l, packed_l, film_l, stats_l, p_in, g_p, g_mu, g_lv, dp_in, \
                               dcanon_l, dfm_l, scratch_a, scratch_b, flow_eps, workspace, (hipStream_t)stream);
    switch (t_ns(precision)) {
        case 2: DPF_BWD(2)
        case 3: DPF_BWD(3)
        default: return DPF_ENOSUP;
    }
#undef DPF_BWD
}
