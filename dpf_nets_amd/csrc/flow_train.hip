// Training-mode conditional affine-coupling layer for gfx950 (MI355X): forward with
// batch-statistics BatchNorm and the full backward pass, one coupling layer per call.
//
// Replaces, for model.train(), CondRealNVPFlow3D.forward (lib/networks/flows.py:95-117) and
// what autograd derives from it (lib/networks/training.py:55).  In training mode the layer is
// NOT a per-point map: both BatchNorm1d layers of each conditioner branch normalise with
// statistics over all B*N points (flows.py:27,30,62,65), which puts two grid-wide reductions in
// the forward pass and two in the backward pass.  Each reduction is a kernel boundary:
//
//  forward   (tstats_x     moments of the (<=2) kept coordinates: first layer of a call only -- afterwards the
//                          previous layer's flow kernel leaves them behind in its epilogue)
//            tstats_h1     prologue (every workgroup): BN0 batch stats as an analytic function of those moments
//                          (h0 = W0 x is linear) -> folded input-MFMA fragments with the BATCH statistics;
//                          h1 = W1 relu(BN0(W0 x)) on the matrix cores, sum / sum of squares per workgroup
//            tfold         those partials -> totals (fixed order, no global atomics) -> BN1 batch stats + this
//                          step's FiLM vectors -> the per-cloud block the eval kernel consumes
//            flow_kernel   (csrc/flow.hip, L = 1) the layer itself (+ the next layer's input moments)
//  backward  tbwd1         prologue: pass 3 of the layer ABOVE (d gamma0, d beta0, dW1, dW0 from its totals -- BN0
//                          backward in closed form: h0n is linear in x, so its sums over points follow from the x
//                          moments -- and the conditioner path of its input gradient, u_k - affine(x), added to the
//                          gradient this layer receives); recompute to h2; d(outputs) -> d(o) (4 floats/point,
//                          stored), dW2, db2, per-cloud d FiLM(a, c): per-workgroup partial rows
//            tbwd2         (17 role workgroups at the front of the grid: rows -> per-cloud totals -> d FiLM, dW2, db2 and
//                          the BN1-backward means, published behind a counter; the others wait for it behind their
//                          weight staging); recompute; BN1
//                          backward; dh0 = W1^T dh1 and dW1 = dh1 h0^T on the matrix cores; per point
//                          u_k = sum_f c_fk dh0a[f] (2 floats, stored); per feature sums of
//                          dh0a * {1, x_a - E x_a, x_b - E x_b}  (sum dh0a * h0n follows from them: h0n is linear in x)
//            tcolsum       per-workgroup partials -> totals
//            (tbwd3f       pass 3 as its own launch: last layer of a call only)
// (every dependent launch costs ~4.5 us on this part however small the kernel, so the tiny finishing steps are
// recomputed by their consumers instead of being kernels of their own: 6 launches per layer, 12 in r01)
//
// Activations are RECOMPUTED from the layer input in both backward passes (MFMA work is cheap);
// nothing of size (B*N, 64) ever goes to HBM -- between the passes travel 4 + 2 floats per point.
// The per-cloud FiLM conditioner nets (B x 64 tensors) are computed outside this file (csrc/film_train.hip: all 4 L of
// them in one launch each way; batched tensor ops for B > 64); they enter here as the tensor `fm` and leave as `dfm`.
//
// Precision: the forward contraction h1 = W1 relu(h0) -- and its recomputation in the backward passes,
// which decides every ReLU mask -- runs at the precision the caller asks for: bf16x3 (hi/lo split,
// ~1e-5) or bf16x6 (hi/mid/lo, fp32-class; the default of the host side, because a ReLU whose
// pre-activation is within the forward error of zero switches the other way and moves that point's
// gradient to the other subgradient: at 1e-5 that happens to ~1e-5 of all ReLUs).  The gradient
// contractions themselves (dh0 = W1^T dh1, dW1 = dh1 h0^T) use hi/lo splits (3 products).
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "flow_common.h"
#include "graph_cache.h"
#include "zero_fill.h"

namespace {

constexpr int TW = 8;                    // waves per workgroup (256 points of one cloud)
constexpr int TBLK = TW * TILE;

// ---- compact per-layer parameter block `tcanon` (floats), branch order (logvar, mu) ----------
constexpr int T_W0 = 0;        // [64][nk] sd0.weight as stored (nk = 1 or 2 keep channels), then zeros up to 128
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
__host__ __device__ constexpr int pt_a0n(int NS) { return pt_a0(NS) + 4096; }        // 4 KiB nobody fills (r01-r04: a second input-MFMA set for the normalised h0); in LDS: pass 2's dh1 fragments
__host__ __device__ constexpr int pt_a1t(int NS) { return pt_a0(NS) + 8192; }        // W1^T fragments, hi | lo
__host__ __device__ constexpr int pt_tail(int NS) { return pt_a1t(NS) + 2 * P_A1_PART; }   // 1 KiB (a whole staging piece): [br] 2^-kw of the fp16 W1^T
// (the two-part precisions only: with three forward parts pass 2 already uses all 160 KiB of LDS, and bf16x6 does not read it)
__host__ __device__ constexpr int pt_bytes(int NS) { return pt_tail(NS) + (NS == 2 ? 1024 : 0); }

// ---- per-layer saved statistics (floats) ---------------------------------------------------------
// stats[br][k][64]: k = 0 mean0, 1 rstd0, 2 mean1, 3 rstd1, 4 batch var0 (unbiased), 5 batch var1 (unbiased)
// then the moments of the kept coordinates: E[xa], E[xb], cov aa, bb, ab
constexpr int ST_BR = 6 * 64;
constexpr int ST_MOM = 2 * ST_BR;
constexpr int ST_LAYER = ST_MOM + 8;

// ---- backward FiLM block per (layer, cloud) (floats): [br][k][64], k = 0 a, 1 c, 2 rstd1, 3 c/a ---
constexpr int FB_BR = 4 * 64;
constexpr int FB_CLOUD = 2 * FB_BR;

// floats of a workgroup's pass-2 partial row per branch (written by tbwd2_kernel).  r05: the sum of dh0a * h0n is no longer a
// column -- h0n = rstd0 (w_a (x_a - E x_a) + w_b (x_b - E x_b)) is linear in x, so it follows from the two CENTRED input sums
// (bwd3_coefs), which also frees them of the cancellation against E x
constexpr int P2_S = 0;          // [64]   sum dh0a                       (d beta0)
constexpr int P2_W = 64;         // [4096] dW1, row = out feature
constexpr int P2_A = 4160;       // [64]   sum dh0a * (x_a - E x_a)
constexpr int P2_B = 4224;       // [64]   sum dh0a * (x_b - E x_b)
constexpr int P2_J = 4288;
static_assert(P2_J % 32 == 0 && P2_W % 32 == 0 && P2_A % 32 == 0, "column workgroups own 32 columns");

__device__ __forceinline__ f32x16 zero16() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}

// ===================================================================================================
// pack: W1 and W1^T fragments for every layer (once per optimizer step)
// ===================================================================================================
template <int NS, bool F16 = false>
__global__ __launch_bounds__(256) void tpack_kernel(const float *__restrict__ tcanon, uint8_t *__restrict__ packed) {
    const int l = blockIdx.x;
    const float *cl = tcanon + (size_t)l * T_LAYER;
    uint16_t *o1 = (uint16_t *)(packed + (size_t)l * pt_bytes(NS) + PT_A1);
    uint16_t *oT = (uint16_t *)(packed + (size_t)l * pt_bytes(NS) + pt_a1t(NS));
    __shared__ float wmax[2][256], wscale[2];
    {   // largest |W1| per branch -> the power-of-two scale of the fp16 W1^T fragments (1 when the branch is all zeros)
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            float m = 0.f;
            for (int i = threadIdx.x; i < 4096; i += 256) m = fmaxf(m, fabsf(cl[br * T_BR + T_W1 + i]));
            wmax[br][threadIdx.x] = m;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            float m = 0.f;
            for (int i = 0; i < 256; ++i) m = fmaxf(m, wmax[threadIdx.x][i]);
            int e = (int)((f2u(m) >> 23) & 0xFFu) - 127;                   // m in [2^e, 2^(e+1))
            const bool ok = m > 0.f && e > -100 && e < 100;
            wscale[threadIdx.x] = ok ? u2f((uint32_t)(127 + 13 - e) << 23) : 1.0f;
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < 2 * 2 * 4 * 64 * 8; idx += blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 3, tp = (idx >> 11) & 1, br = idx >> 12;
        const int i = lane & 31, h = lane >> 5;
        const int fk = acc_feature(s >> 1, 8 * (s & 1) + j, h);     // feature carried by K slot (s, j, h)
        const float *W1 = cl + br * T_BR + T_W1;
        float r1, r2;
        const float w = W1[(32 * tp + i) * 64 + fk];                // forward: rows = out feature, K = in feature
        if (F16) {                     // fp16 hi (RNE) + fp16 of the exact remainder of W1 * 2^kw, as csrc/flow.hip pack_kernel<2, true> (r05: the
            // forward fragments carry the same power of two as the W1^T fragments below; tfold_kernel puts D 2^kw and W2' 2^-kw
            // into the layer's FiLM block, the statistics and pass 1 unscale their sums of the pre-activation, pass 2 its C1)
            const float wsf = w * wscale[br];
            const _Float16 wh = (_Float16)wsf;
            const _Float16 wl = (_Float16)(wsf - (float)wh);
            o1[idx] = __builtin_bit_cast(uint16_t, wh);
            o1[P_A1_PART / 2 + idx] = __builtin_bit_cast(uint16_t, wl);
        } else if (NS == 2) {
            o1[idx] = (uint16_t)(split_hi(w, r1) >> 16);
            o1[P_A1_PART / 2 + idx] = (uint16_t)bf16_rne(r1);
        } else {
            o1[idx] = (uint16_t)(split_hi(w, r1) >> 16);
            o1[P_A1_PART / 2 + idx] = (uint16_t)(split_hi(r1, r2) >> 16);
            o1[P_A1_PART + idx] = (uint16_t)bf16_rne(r2);
        }
        const float wt = W1[fk * 64 + (32 * tp + i)];               // transposed: rows = in feature, K = out feature
        if (F16) {
            // r04: the gradient contractions of an f16x3 stack take fp16 hi + fp16 lo operands (22 significant bits instead of
            // the 16 of a bf16 hi/lo pair).  W1 is scaled by a power of two per (layer, branch) so that its largest entry sits in
            // [2^13, 2^14): hi AND lo are normal fp16 numbers for every entry within 2^-11 of the largest (at the init scale
            // of 0.01 the unscaled remainders were fp16 subnormals: ~17 bits); the consumer multiplies by 2^-kw (exact).
            const float ws = wt * wscale[br];
            const _Float16 wh = (_Float16)ws;
            const _Float16 wl = (_Float16)(ws - (float)wh);
            oT[idx] = __builtin_bit_cast(uint16_t, wh);
            oT[P_A1_PART / 2 + idx] = __builtin_bit_cast(uint16_t, wl);
        } else {
            oT[idx] = (uint16_t)(split_hi(wt, r1) >> 16);
            oT[P_A1_PART / 2 + idx] = (uint16_t)bf16_rne(r1);
        }
    }
    if (NS == 2 && threadIdx.x < 2) ((float *)(packed + (size_t)l * pt_bytes(NS) + pt_tail(NS)))[threadIdx.x] = 1.0f / wscale[threadIdx.x];
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

// BN0 batch statistics (analytic: h0 = W0 x is linear in x) and the folded input-MFMA fragments of the layer.
// One workgroup, thread = (branch, feature).
// 512 threads: the moment rows take all of them, the per-feature folds 128, the 2048 fragment slots 256
// (with 128 threads the slot loop alone was 16 serial iterations of ~80 instructions: 9.3 us for the kernel)
// (r02: a device function run by every workgroup of tstats_h1 -- the consumer -- straight into its LDS: 2 048 partial
// values and 8 KiB of fragments are cheaper to redo 256 times than a dependent one-workgroup launch is to wait for;
// workgroup 0 also publishes the fragments and the statistics for the kernels that follow.  512 threads.)
#ifdef DPF_PROFILE
__device__ unsigned long long *g_kprof = nullptr;            // [kernel id][8] s_memtime stamps of workgroup 0, wave 0
#define KP(kid, i) { __builtin_amdgcn_sched_barrier(0); if (g_kprof != nullptr && blockIdx.x == 1 && blockIdx.y == 3 && threadIdx.x == 0) g_kprof[(kid) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define KPR(i) { __builtin_amdgcn_sched_barrier(0); if (g_kprof != nullptr && id == 1 && threadIdx.x == 0) g_kprof[6 * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define KP(kid, i) {}
#define KPR(i) {}
#endif

// a workgroup barrier that orders LDS traffic only (see stage_load below)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// the loads of bn0_fold, to be issued BEFORE the caller's weight loads (vector-memory results return in order: issued behind
// 48+ KB of weights they wait for all of it): this thread's share of the moment partials, and (threads < 128) its feature's
// W0 / gamma / beta
struct Bn0Loads { double ld[4][5]; double v[5]; float wa, wb, gamma, beta; };
__device__ __forceinline__ Bn0Loads bn0_loads(int nblk, int nk, const double *__restrict__ part, const float *__restrict__ tcanon_l) {
    Bn0Loads L;
    // the first 4 x 512 rows (all of them up to B * N / 256 = 2048): requested here, CONSUMED in bn0_fold.  r03: summing them
    // here, inside a run-time loop over row blocks, had put an `s_waitcnt vmcnt(0)` in front of the caller's weight staging --
    // the cold round trip of these loads and the weight staging ran one after the other
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = threadIdx.x + k * TW * 64;
        const bool ok = row < nblk;
#pragma unroll
        for (int i = 0; i < 5; ++i) L.ld[k][i] = ok ? part[(size_t)(ok ? row : 0) * 8 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) L.v[i] = 0;
    for (int b0 = 4 * TW * 64; b0 < nblk; b0 += 4 * TW * 64) {             // more rows than that (rare): summed here, after the first block in bn0_fold
        double ld[4][5];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = b0 + threadIdx.x + k * TW * 64;
            const bool ok = row < nblk;
#pragma unroll
            for (int i = 0; i < 5; ++i) ld[k][i] = ok ? part[(size_t)(ok ? row : 0) * 8 + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 5; ++i) L.v[i] += ld[k][i];
    }
    const int q = threadIdx.x & 127, br = q >> 6, f = q & 63;
    const float *cb = tcanon_l + br * T_BR;
    L.wa = cb[T_W0 + f * nk]; L.wb = nk == 2 ? cb[T_W0 + f * 2 + 1] : 0.f; L.gamma = cb[T_G0 + f]; L.beta = cb[T_B0 + f];
    return L;
}

__device__ __forceinline__ void bn0_fold(const Bn0Loads &L, double count, uint8_t *lds_a0, uint8_t *__restrict__ packed_a0,
                                         float *__restrict__ stats_l) {
    // r03: this prologue was HALF of tstats_h1 (11.8 K of 23.6 K ticks, tools/train_kprof.py): two dependent rounds of row
    // loads on 128 threads issued behind the weight DMA, a 16-iteration slot loop of ~80 instructions.  Now the loads are in
    // flight before the DMA on all 512 threads (bn0_loads), and a thread builds the eight K slots of one (branch, tile,
    // lane) -- one 16-byte store.
    __shared__ double mom[5], wsum[TW][5];
    __shared__ float fold[2][64][4];
    {   // fixed-order sum of the per-workgroup partials: thread -> rows tid, tid + 512, ...; wave butterflies; waves in order
        double v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = (((L.ld[0][i] + L.ld[1][i]) + L.ld[2][i]) + L.ld[3][i]) + L.v[i];   // rows tid, tid + 512, ... in order
#pragma unroll
        for (int i = 0; i < 5; ++i)
            for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
        KP(0, 5)
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 5; ++i) wsum[threadIdx.x >> 6][i] = v[i];
        lds_barrier();
        KP(0, 6)
        if (threadIdx.x < 5) {
            const int i = threadIdx.x;
            mom[i] = (((wsum[0][i] + wsum[1][i]) + (wsum[2][i] + wsum[3][i])) + ((wsum[4][i] + wsum[5][i]) + (wsum[6][i] + wsum[7][i]))) / count;
        }
        lds_barrier();
    }
    if (threadIdx.x < 128) {
    const int br = threadIdx.x >> 6, f = threadIdx.x & 63;
    const double wa = L.wa, wb = L.wb;
    const double ea = mom[0], eb = mom[1];
    const double caa = mom[2] - ea * ea, cbb = mom[3] - eb * eb, cab = mom[4] - ea * eb;
    if (threadIdx.x == 0 && stats_l != nullptr) {
        float *m = stats_l + ST_MOM;
        m[0] = (float)ea; m[1] = (float)eb; m[2] = (float)caa; m[3] = (float)cbb; m[4] = (float)cab;
    }
    const double mean = wa * ea + wb * eb;
    double var = wa * wa * caa + wb * wb * cbb + 2.0 * wa * wb * cab;     // biased, as BatchNorm normalises with
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const float gamma = L.gamma, beta = L.beta;
    if (stats_l != nullptr) {
    float *st = stats_l + br * ST_BR;
    st[0 * 64 + f] = (float)mean;
    st[1 * 64 + f] = rstd;
    st[4 * 64 + f] = (float)(var * (count / (count > 1 ? count - 1 : 1)));      // unbiased: running_var update
    }
    const float s0 = gamma * rstd;
    fold[br][f][0] = s0 * (float)wa; fold[br][f][1] = s0 * (float)wb; fold[br][f][2] = beta - (float)mean * s0;
    }
    KP(0, 7)
    lds_barrier();
    if (threadIdx.x < 256) {               // one thread per (branch, tile, lane)
        const int b2 = threadIdx.x >> 7, t = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
        const int ff = 32 * t + (lane & 31), h = lane >> 5;
        const u32x4 v = input_weight_slots8(fold[b2][ff][h], fold[b2][ff][2], h);
        const int off = ((b2 * 2 + t) * 64 + lane) * 16;
        *(u32x4 *)(lds_a0 + off) = v;
        if (packed_a0 != nullptr) *(u32x4 *)(packed_a0 + off) = v;
    }
}

// column sums of a (nrows, J) fp32 matrix of per-workgroup partials, in a fixed order, as doubles.
// workgroup = 32 columns x 32 row groups
// dcanon_l (optional): the dW1 totals (columns P2_W .. P2_W + 4095 of each branch's P2_J) go straight to the layer's gradient block
// as floats (r03: formerly a grid-stride loop in the prologue of every workgroup of the next pass 1)
__global__ __launch_bounds__(1024) void tcolsum_kernel(int nrows, int J, const float *__restrict__ part, double *__restrict__ out,
                                                       float *__restrict__ dcanon_l, int p2j) {
    __shared__ double acc[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    double s = 0;
    if (j < J) {
        int r = rg;
        for (; r + 224 < nrows; r += 256) {                                // eight rows in flight (one round of loads at 256 rows), added in row order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(r + 32 * k) * J + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; r < nrows; r += 32) s += part[(size_t)r * J + j];
    }
    acc[rg][c] = s;
    __syncthreads();
    if (rg == 0 && j < J) {
        double t = 0;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += acc[r][c];
        out[j] = t;
        if (dcanon_l != nullptr) {
            const int b2 = j / p2j, jj = j - b2 * p2j;
            if (jj >= P2_W && jj < P2_W + 4096) dcanon_l[b2 * T_BR + T_W1 + jj - P2_W] = (float)t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// shared tile machinery (one 32-point tile per wave, weights staged in LDS)
// ---------------------------------------------------------------------------------------------------
// Staging a layer's packed block into LDS THROUGH REGISTERS, in two halves: stage_load issues a wave's 16-byte loads (a
// compile-time number of them; a piece index past the end is clamped to the last piece: the same bytes to the same place),
// stage_store writes them to LDS.  r03: these prologues used LDS-DMA (global_load_lds).  The compiler counts LDS-DMA and
// ordinary loads as different kinds of vector-memory events, so once both were in flight EVERY wait became
// `s_waitcnt vmcnt(0)`, and any LDS access behind a DMA waited for it as a possible alias: the small reductions of the
// prologues (BN0 fold, pass-3 coefficients, BN1-backward means) ran only after the whole weight block had landed, and in
// tstats_h1 the weight DMA was not even issued until the moment rows were back.  With plain loads everything is one
// in-order stream: the compiler waits for exactly the loads a value needs, the weights land while the reductions run.
// (tools/ubench/stage_rate.hip: load + ds_write is as fast as the DMA.)  lds_barrier: a workgroup barrier that orders LDS
// traffic only -- __syncthreads() would wait for every outstanding load.
template <int NBYTES>
struct StageRegs {
    static constexpr int NPIECES = (NBYTES + 1023) / 1024, PER_WAVE = (NPIECES + TW - 1) / TW;
    u32x4 v[PER_WAVE];
};
template <int NBYTES>
__device__ __forceinline__ StageRegs<NBYTES> stage_load(const uint8_t *src, int wave, int lane) {
    using R = StageRegs<NBYTES>;
    R r;
#pragma unroll
    for (int i = 0; i < R::PER_WAVE; ++i) {
        int c = wave + TW * i;
        if (TW * i + TW > R::NPIECES) c = c < R::NPIECES ? c : R::NPIECES - 1;
        r.v[i] = *(const u32x4 *)(src + c * 1024 + lane * 16);
    }
    return r;
}
template <int NBYTES>
__device__ __forceinline__ void stage_store(const StageRegs<NBYTES> &r, uint8_t *lds, int wave, int lane) {
    using R = StageRegs<NBYTES>;
#pragma unroll
    for (int i = 0; i < R::PER_WAVE; ++i) {
        int c = wave + TW * i;
        if (TW * i + TW > R::NPIECES) c = c < R::NPIECES ? c : R::NPIECES - 1;
        *(u32x4 *)(lds + c * 1024 + lane * 16) = r.v[i];
    }
}
// input MFMA of one branch: acc[t] = A0[br][t] . b0   (t = M tile)
__device__ __forceinline__ void input_mfma(const uint8_t *a0, int br, int lane, u32x4 b0, f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
        acc[t] = mfma(*(const u32x4 *)(a0 + ((br * 2 + t) * 64 + lane) * 16), b0, zero16());
}

// the same with the operands swapped: acc[t] holds, in LANE pl, feature 32 t + pl of the 16 POINTS (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ void input_mfma_swapped(const uint8_t *a0, int br, int lane, u32x4 b0, f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
        acc[t] = mfma(b0, *(const u32x4 *)(a0 + ((br * 2 + t) * 64 + lane) * 16), zero16());
}

// Swapped accumulators (lane = feature, registers = points) as bf16 hi/lo operand fragments whose K dimension is the
// tile's 32 points: k-step j takes registers 8 j .. 8 j + 7 of the lane, i.e. k-slot i of lane-half kg holds point
// (i & 3) + 8 (2 j + (i >> 2)) + 4 kg -- some fixed order of the k-step's 16 points, the same for every operand built here.
// Symmetric hi/lo split of a pair: hi = bf16 round-to-nearest-even, lo = bf16(x - hi), signed.  The truncating split of the
// forward path (split_hi: the remainder has the sign of x) leaves a lo.lo term of the sign of the product out of every
// hi.hi + hi.lo + lo.hi contraction -- a bias of ~2^-18 per term that does not average out over a sum of 16 384 terms
// (r02: the cancelling bias gradients of the deeper layers).  With a zero-mean remainder on the activation side the omitted
// term is zero-mean.  Same six instructions per pair.  Backward-only operands: the forward recomputation keeps the forward
// kernel's split, bit for bit.
__device__ __forceinline__ void split_pair_sym(float v0, float v1, uint32_t &hi, uint32_t &lo) {
    hi = pack_bf16_rne(v0, v1);
    lo = pack_bf16_rne(v0 - u2f(hi << 16), v1 - u2f(hi & 0xFFFF0000u));
}

// the same for fp16 hi/lo operands (r04: the gradient contractions of an f16x3 stack): hi = fp16(x) and lo = fp16(x - hi), both
// round-to-nearest-even -- one v_cvt_pk_f16_f32 and the two v_fma_mix*_f16 of split_relu_f16 (`negone`: see there).  x must
// be scaled into fp16's range by the caller; 11 + 11 bits for |x| >= 2^-3, the absolute 2^-24 of fp16's subnormals below.
__device__ __forceinline__ void split_pair_sym_f16(float v0, float v1, float negone, uint32_t &hi, uint32_t &lo) {
    const f32x2 v = {v0, v1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f16x2 l = {(_Float16)__builtin_fmaf((float)h[0], negone, v0), (_Float16)__builtin_fmaf((float)h[1], negone, v1)};
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
template <bool RELU, bool SCALE = true>
__device__ __forceinline__ void kfrags_from_swapped_f16(const f32x16 (&v)[2], float scale, float negone, u32x4 (&hi)[2][2], u32x4 (&lo)[2][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float v0 = RELU ? relu(v[t][8 * j2 + 2 * d]) : v[t][8 * j2 + 2 * d];
                float v1 = RELU ? relu(v[t][8 * j2 + 2 * d + 1]) : v[t][8 * j2 + 2 * d + 1];
                if constexpr (SCALE) { v0 *= scale; v1 *= scale; }
                uint32_t hp, lp;
                split_pair_sym_f16(v0, v1, negone, hp, lp);
                hi[t][j2][d] = hp;
                lo[t][j2][d] = lp;
            }
}

template <bool RELU>
__device__ __forceinline__ void kfrags_from_swapped(const f32x16 (&v)[2], u32x4 (&hi)[2][2], u32x4 (&lo)[2][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float v0 = RELU ? relu(v[t][8 * j2 + 2 * d]) : v[t][8 * j2 + 2 * d];
                const float v1 = RELU ? relu(v[t][8 * j2 + 2 * d + 1]) : v[t][8 * j2 + 2 * d + 1];
                uint32_t hp, lp;
                split_pair_sym(v0, v1, hp, lp);
                hi[t][j2][d] = hp;
                lo[t][j2][d] = lp;
            }
}

// bf16 split (NS parts) of an accumulator fragment pair into the B fragments of the next contraction
// (register r of M tile t = element j = r&7 of k-step 2t + (r>>3)); RELU = clamp at zero first.
// Same arithmetic as branch_tile in csrc/flow.hip: the recomputed activations are bit-identical to
// the forward kernel's.
template <bool RELU, int NS, bool SYM = false, bool F16 = false>
__device__ __forceinline__ void split_fragment(const f32x16 (&v)[2], u32x4 (&bf)[NS][4], float negone = -1.0f) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;
            if constexpr (F16) {               // the forward kernel's fp16 hi/lo split with the ReLU folded in (flow_common.h)
                static_assert(!F16 || (NS == 2 && RELU && !SYM), "f16x3: forward operands only");
                uint32_t hp, lp;
                split_relu_f16(v[t][r], v[t][r + 1], negone, hp, lp);
                bf[0][s][d] = hp;
                bf[1][s][d] = lp;
                continue;
            }
            const float v0 = RELU ? relu(v[t][r]) : v[t][r], v1 = RELU ? relu(v[t][r + 1]) : v[t][r + 1];
            if constexpr (SYM) {               // backward-only operand (NS == 2): symmetric split
                static_assert(!SYM || NS == 2, "symmetric split: hi/lo only");
                uint32_t hp, lp;
                split_pair_sym(v0, v1, hp, lp);
                bf[0][s][d] = hp;
                bf[NS - 1][s][d] = lp;
                continue;
            }
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

// acc[tp] += A1[br] . B  with the split terms of Terms<NS>; a1 = base of [part][br][tp][s][lane].  Same order as
// branch_tile in csrc/flow.hip (k-step major; the fragment of each part is loaded once and feeds every term that
// uses it): the recomputed pre-activations are bit-identical to the forward kernel's.
template <int NS, bool F16 = false>
__device__ __forceinline__ void chain_mfma(const uint8_t *a1, int br, int lane, const u32x4 (&bf)[NS][4], f32x16 (&acc)[2]) {
    using TT = Terms<NS>;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 af[NS][2];
#pragma unroll
        for (int part = 0; part < NS; ++part)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                af[part][tp] = *(const u32x4 *)(a1 + part * P_A1_PART + (((br * 2 + tp) * 4 + s) * 64 + lane) * 16);
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                acc[tp] = F16 ? mfma_f16(af[TT::A[term]][tp], bf[TT::B[term]][s], acc[tp]) : mfma(af[TT::A[term]][tp], bf[TT::B[term]][s], acc[tp]);
    }
}

// The same contraction with the operands swapped (activations = A, weights = B; the two fragment layouts of the
// 32x32x16 MFMA mirror each other, so the same registers and the same packed fragments serve): the accumulator of
// M tile tp then holds, in LANE pl, feature 32*tp + pl of the 16 POINTS (r&3) + 8*(r>>2) + 4h of the tile -- sums
// over the points of a tile become in-lane adds plus one cross-half swap instead of a cross-lane butterfly.
template <int NS, bool F16 = false>
__device__ __forceinline__ void chain_mfma_swapped(const uint8_t *a1, int br, int lane, const u32x4 (&bf)[NS][4], f32x16 (&acc)[2]) {
    using TT = Terms<NS>;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 af[NS][2];
#pragma unroll
        for (int part = 0; part < NS; ++part)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                af[part][tp] = *(const u32x4 *)(a1 + part * P_A1_PART + (((br * 2 + tp) * 4 + s) * 64 + lane) * 16);
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                acc[tp] = F16 ? mfma_f16(bf[TT::B[term]][s], af[TT::A[term]][tp], acc[tp]) : mfma(bf[TT::B[term]][s], af[TT::A[term]][tp], acc[tp]);
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
__device__ __forceinline__ void reduce_stage(float (&w)[16], int pl) {
    constexpr int n2 = 8 >> (K - 1);
    const bool up = (pl >> K) & 1;
#pragma unroll
    for (int i = 0; i < n2; ++i) {
        const float keep = up ? w[i + n2] : w[i];
        const float send = up ? w[i] : w[i + n2];
        w[i] = keep + __shfl_xor(send, 1 << K);
    }
}
// gen(i), i = 0..31: element of accumulator register i (tile i >> 4, register i & 15); elements are
// produced on the fly so that a product like dh * x never exists as 32 live registers
template <typename Gen>
__device__ __forceinline__ float reduce_points_gen(Gen gen, int pl) {
    float w[16];
    const bool up = pl & 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float lo = gen(i), hi = gen(i + 16);
        const float keep = up ? hi : lo, send = up ? lo : hi;
        w[i] = keep + __shfl_xor(send, 1);
    }
    reduce_stage<1>(w, pl); reduce_stage<2>(w, pl); reduce_stage<3>(w, pl); reduce_stage<4>(w, pl);
    __builtin_amdgcn_sched_barrier(0);       // keep consecutive reductions from interleaving (register pressure)
    return w[0];
}
__device__ __forceinline__ float reduce_points(const f32x16 (&v)[2], int pl) {
    return reduce_points_gen([&](int i) { return v[i >> 4][i & 15]; }, pl);
}
__device__ __forceinline__ int reduced_feature(int pl, int h) {
    const int R = ((pl & 1) << 4) | ((pl & 2) << 2) | (pl & 4) | ((pl & 8) >> 2) | ((pl & 16) >> 4);
    return acc_feature(R >> 4, R & 15, h);
}


struct TArgs {
    const uint8_t *packed_l;     // pt_bytes(NS) of this layer
    const float *tcanon_l;       // T_LAYER floats
    const float *film_l;         // (B, 512) eval FiLM blocks of this layer
    const float *filmb_l;        // (B, FB_CLOUD) backward FiLM blocks
    const float *stats_l;        // ST_LAYER
    const float *p_in;           // (B, 3, N)
    int B, N, ka, kb, wa, wb, mode;
    float eps;
    float negone;                // -1.0f, as a kernel argument so that it sits in an SGPR the compiler cannot see through (split_relu_f16)
};

// h1 = W1 relu(BN0(W0 x)) for every point; per-workgroup partial sums and sums of squares
//   part[blk][br][2][64]
// (gridDim.z == 2, small batches -- at most half a workgroup per CU: one conditioner branch per workgroup, as tbwd2_kernel)
template <int NS, bool F16 = false>
__global__ __launch_bounds__(TW * 64) void tstats_h1_kernel(TArgs a, float *__restrict__ part, int nblk_x, double count,
                                                            const double *__restrict__ xpart, uint8_t *__restrict__ packed_a0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ float acc_s[TW][2][2][64];                                 // per-wave slots (LDS float atomics are slow)
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KP(0, 0)
    // every small load first, then the weight loads (results return in order)
    Bn0Loads bl = bn0_loads(nblk_x, a.kb >= 0 ? 2 : 1, xpart, a.tcanon_l);
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const float *pc = a.p_in + (size_t)bi * 3 * N;
    const int nc = valid ? n : N - 1;
    const float xa = pc[(size_t)a.ka * N + nc], xb = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
    asm volatile("" ::: "memory");
    // f16x3: 2^-kw of each branch's packed W1 (the sums below are of 2^kw h1)
    const float wsinv0 = F16 ? *(const float *)(a.packed_l + pt_tail(NS)) : 1.0f, wsinv1 = F16 ? *(const float *)(a.packed_l + pt_tail(NS) + 4) : 1.0f;
    const StageRegs<NS * P_A1_PART> wregs = stage_load<NS * P_A1_PART>(a.packed_l, wave, lane);   // A1; the A0 fragments are folded right here
#pragma unroll
    for (int k = 0; k < 4; ++k)                                           // pin the consumers of the moment rows BEHIND the weight loads' issue
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(bl.ld[k][i]));
    {
        const bool first = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
        bn0_fold(bl, count, smem + pt_a0(NS), first ? packed_a0 : nullptr, first ? const_cast<float *>(a.stats_l) : nullptr);
    }
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    stage_store(wregs, smem, wave, lane);
    KP(0, 1)
    __syncthreads();
    KP(0, 2)
    const bool split = gridDim.z == 2;
    const int br_lo = split ? (int)blockIdx.z : 0, br_hi = split ? (int)blockIdx.z + 1 : 2;
    for (int br = br_lo; br < br_hi; ++br) {
        f32x16 acc0[2], acc1[2] = {zero16(), zero16()};
        u32x4 bf[NS][4];
        input_mfma(smem + pt_a0(NS), br, lane, b0, acc0);
        split_fragment<true, NS, false, F16>(acc0, bf, a.negone);
        chain_mfma_swapped<NS, F16>(smem + PT_A1, br, lane, bf, acc1);         // lane = feature, registers = points
        const int tile0 = (blockIdx.x * TW + wave) * TILE;
        if (tile0 + TILE > N) {                  // ragged last tile of a cloud (wave-uniform, rare): padding points do not count
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[t][r] = tile0 + (r & 3) + 8 * (r >> 2) + 4 * h < N ? acc1[t][r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc1[t][r];
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if constexpr (F16) { const float u = br ? wsinv1 : wsinv0; s1 *= u; s2 *= u * u; }      // back to h1's scale (exact)
            if (!h) {
                acc_s[wave][br][0][32 * t + pl] = s1;
                acc_s[wave][br][1][32 * t + pl] = s2;
            }
        }
    }
    KP(0, 3)
    __syncthreads();
    if (threadIdx.x < 256 && (int)(threadIdx.x >> 7) >= br_lo && (int)(threadIdx.x >> 7) < br_hi) {     // this workgroup's branch(es) of the row
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < TW; ++w) t += ((const float *)acc_s)[w * 256 + threadIdx.x];
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = t;
    }
    KP(0, 4)
}

// BN1 batch statistics from tstats_h1's per-workgroup partials, then this step's FiLM fold:
//   eval block  (csrc/flow.hip film layout): D = FC/FA, W2' = W2*FA, b2
//   bwd block   a = eps + e^cw, c = cb, rstd1, c/a
// part: (nrows, [br][2][64]) floats;  fm_l: [br][sub(w,b)][B][64]
// One launch instead of a column-sum launch + a fold launch (a dependent tiny launch costs ~4.5 us): workgroup k owns 16
// features of one branch -- its 32 columns (sum, sum of squares) are summed exactly as tcolsum_kernel does (32 row groups,
// then the groups in order, doubles), then its threads fold (cloud, feature) items for all B clouds.
// wtail (f16x3; else NULL): [br] 2^-kw of the layer's packed W1 -- the eval block then carries D 2^kw and W2' 2^-kw, and the sums of
// tstats_h1 arrive already unscaled
__global__ __launch_bounds__(1024) void tfold_kernel(double count, int nrows, const float *__restrict__ part,
                                                     const float *__restrict__ tcanon_l, const float *__restrict__ fm_l,
                                                     int B, float eps, float *__restrict__ stats_l,
                                                     float *__restrict__ film_l, float *__restrict__ filmb_l,
                                                     const float *__restrict__ wtail) {
    __shared__ double acc[32][33];
    __shared__ double tot[32];
    const int br = blockIdx.x >> 2, f0 = (blockIdx.x & 3) * 16;
    const float *cbp = tcanon_l + br * T_BR;
    // r03: the fold's own operands (this step's FiLM vectors, W2 rows) do not depend on the column sums: requested first, so
    // that the kernel is ONE global round trip deep instead of two
    float pre_cw = 0.f, pre_cb = 0.f, pre_w2a = 0.f, pre_w2b = 0.f;
    const float wsinv = wtail != nullptr ? wtail[br] : 1.0f, wsc = 1.0f / wsinv;      // (powers of two: exact)
    if ((int)threadIdx.x < B * 16) {
        const int b = threadIdx.x >> 4, f = f0 + (threadIdx.x & 15);
        pre_cw = fm_l[((size_t)(br * 2 + 0) * B + b) * 64 + f]; pre_cb = fm_l[((size_t)(br * 2 + 1) * B + b) * 64 + f];
        pre_w2a = cbp[T_W2 + f]; pre_w2b = cbp[T_W2 + 64 + f];
    }
    {
        const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
        const int col = (br * 2 + (c >> 4)) * 64 + f0 + (c & 15);
        double s = 0;
        int r = rg;
        for (; r + 224 < nrows; r += 256) {                                // eight rows in flight, added in row order (r03)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(r + 32 * k) * 256 + col];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; r < nrows; r += 32) s += part[(size_t)r * 256 + col];
        acc[rg][c] = s;
        __syncthreads();
        if (rg == 0) {
            double t = 0;
#pragma unroll
            for (int r2 = 0; r2 < 32; ++r2) t += acc[r2][c];
            tot[c] = t;
        }
        __syncthreads();
    }
    for (int item = threadIdx.x; item < B * 16; item += 1024) {
        const int b = item >> 4, fl = item & 15, f = f0 + fl;
        const double mean = tot[fl] / count;
        double var = tot[16 + fl] / count - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
        if (b == 0) {
            float *st = stats_l + br * ST_BR;
            st[2 * 64 + f] = (float)mean;
            st[3 * 64 + f] = rstd;
            st[5 * 64 + f] = (float)(var * (count / (count > 1 ? count - 1 : 1)));
        }
        const bool first = item == (int)threadIdx.x;
        const float cw = first ? pre_cw : fm_l[((size_t)(br * 2 + 0) * B + b) * 64 + f];
        const float cb = first ? pre_cb : fm_l[((size_t)(br * 2 + 1) * B + b) * 64 + f];
        const float w2a_v = first ? pre_w2a : cbp[T_W2 + f], w2b_v = first ? pre_w2b : cbp[T_W2 + 64 + f];
        const float av = eps + expf(cw);
        const float FA = av * rstd, FC = -av * (float)mean * rstd + cb;
        float *o = film_l + (size_t)b * (FILM_BYTES / 4) + br * FILM_BR_FLOATS;
        o[f] = (FC / FA) * wsc;
        o[64 + f] = (w2a_v * FA) * wsinv;
        o[128 + f] = (w2b_v * FA) * wsinv;
        if (f < 2) film_l[(size_t)b * (FILM_BYTES / 4) + FILM_B2_OFF + br * 2 + f] = cbp[T_B2 + f];
        float *ob = filmb_l + (size_t)b * FB_CLOUD + br * FB_BR;
        ob[0 * 64 + f] = av;
        ob[1 * 64 + f] = cb;
        ob[2 * 64 + f] = rstd;
        ob[3 * 64 + f] = cb / av;
    }
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

// d(o) of one branch from d(h2a): the per-feature part shared by passes 1 and 2 (same arithmetic in
// both, so the sums of pass 1 describe exactly the dh2a pass 2 works with)
__device__ __forceinline__ float dh2a_of(float pre, float w2a, float w2b, float doa, float dob) {
    return pre > 0.f ? w2a * doa + w2b * dob : 0.f;
}

// Totals of pass 2 of a layer -> the coefficients of the conditioner path of its input gradient,
//   dx_k[pt] = u_k[pt] - coef[4k] - coef[4k+1] (x_a - E x_a) - coef[4k+2] (x_b - E x_b)   (see tbwd3f_kernel; coef[3], coef[7] = E x_a, E x_b),
// left in LDS (`coef`, 8 floats; `acc` is 8 KiB of scratch); EVERY workgroup of the caller runs this (128 threads of double
// arithmetic, same operations in the same order, so the same bits), workgroup 0 also writes d gamma0 / d beta0 / dW0 and
// the dW1 totals are converted by whoever comes first (grid-stride).  Ends with a workgroup barrier.
struct CoefLoads { double S, Sa, Sb; float ea, eb, caa, cbb, cab, wa, wb, gamma, rstd0; };
// the loads of bwd3_coefs (threads < 128), to be issued BEFORE the caller's weight loads
// (two halves: what does not come from the column sums -- the layer's statistics and weights -- and the three totals)
__device__ __forceinline__ void bwd3_loads_const(CoefLoads &L, int nk, const float *__restrict__ tcanon_l, const float *__restrict__ stats_l) {
    const int q = threadIdx.x & 127, br = q >> 6, f = q & 63;
    const float *cb = tcanon_l + br * T_BR;
    const float *m = stats_l + ST_MOM;
    L.ea = m[0]; L.eb = m[1]; L.caa = m[2]; L.cbb = m[3]; L.cab = m[4];
    L.wa = cb[T_W0 + f * nk]; L.wb = nk == 2 ? cb[T_W0 + f * 2 + 1] : 0.f; L.gamma = cb[T_G0 + f];
    L.rstd0 = stats_l[br * ST_BR + 64 + f];
}
template <bool AGENT = false>
__device__ __forceinline__ void bwd3_loads_totals(CoefLoads &L, const double *__restrict__ tot) {
    const int q = threadIdx.x & 127, br = q >> 6, f = q & 63;
    const double *t = tot + (size_t)br * P2_J;
    if (AGENT) {       // written by other workgroups of THIS launch (colsum_role): past this CU's L1 and the XCD's L2
        L.S = __hip_atomic_load(&t[P2_S + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.Sa = __hip_atomic_load(&t[P2_A + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.Sb = __hip_atomic_load(&t[P2_B + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        L.S = t[P2_S + f]; L.Sa = t[P2_A + f]; L.Sb = t[P2_B + f];
    }
}
template <bool AGENT = false>
__device__ __forceinline__ CoefLoads bwd3_loads(int nk, const double *__restrict__ tot, const float *__restrict__ tcanon_l,
                                                const float *__restrict__ stats_l) {
    CoefLoads L;
    bwd3_loads_totals<AGENT>(L, tot);
    bwd3_loads_const(L, nk, tcanon_l, stats_l);
    return L;
}
__device__ __forceinline__ void bwd3_coefs(const CoefLoads &L, int blk, int nk, double count, float *__restrict__ dcanon_l,
                                           double (*acc)[8], float *coef) {
    // r03: 8.9 K of tbwd1's 31 K ticks were this prologue (tools/train_kprof.py): loads issued behind the weight DMA, eight
    // serial 128-term double sums out of LDS by eight threads, three barriers, and a grid-stride dW1 conversion loop.  Now:
    // loads in flight before the DMA (bwd3_loads), wave butterflies, two partials per value, one barrier; tcolsum converts
    // dW1 itself.  acc: [2][8] doubles of scratch.
    if (threadIdx.x < 128) {
        const int br = threadIdx.x >> 6, f = threadIdx.x & 63;
        const double S = L.S, Sa = L.Sa, Sb = L.Sb;                                                // Sa, Sb: against the CENTRED inputs
        const double caa = L.caa, cbb = L.cbb, cab = L.cab;
        const double wa = L.wa, wb = L.wb, gamma = L.gamma;
        const double rstd0 = L.rstd0;
        const double Sg = rstd0 * (wa * Sa + wb * Sb);                                             // sum dh0a * h0n
        const double A = S / count, Bc = Sg / count;
        const double hxa = rstd0 * (wa * caa + wb * cab), hxb = rstd0 * (wa * cab + wb * cbb);     // E[h0n x_k]
        const double sc = rstd0 * gamma;
        if (blk == 0) {
            dcanon_l[br * T_BR + T_G0 + f] = (float)Sg;
            dcanon_l[br * T_BR + T_B0 + f] = (float)S;
            dcanon_l[br * T_BR + T_W0 + f * nk] = (float)(sc * (Sa - Bc * hxa * count));          // (Sa + E x_a S) - A E x_a P = Sa
            if (nk == 2) dcanon_l[br * T_BR + T_W0 + f * 2 + 1] = (float)(sc * (Sb - Bc * hxb * count));
            else dcanon_l[br * T_BR + T_W0 + 64 + f] = 0.f;
        }
        const double ck[2] = {wa * sc, wb * sc};
        double v[6];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v[3 * k + 0] = ck[k] * A;
            v[3 * k + 1] = ck[k] * Bc * rstd0 * wa;
            v[3 * k + 2] = ck[k] * Bc * rstd0 * wb;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
            for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
        if (f == 0)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[br][i] = v[i];
    }
    lds_barrier();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        coef[k * 4 + 0] = (float)(acc[0][3 * k + 0] + acc[1][3 * k + 0]);
        coef[k * 4 + 1] = (float)(acc[0][3 * k + 1] + acc[1][3 * k + 1]);
        coef[k * 4 + 2] = (float)(acc[0][3 * k + 2] + acc[1][3 * k + 2]);
        coef[k * 4 + 3] = k ? L.eb : L.ea;                                  // the batch means the consumer centres the inputs with
    }
    lds_barrier();
}

// u_k - C_k - alpha_k x_a - beta_k x_b, every operation rounded on its own (no contraction: pass 3 as its own launch
// and pass 3 inside the next layer's pass 1 must agree bit for bit)
__device__ __forceinline__ float cond_path(float u, float c0, float c1, float c2, float xa, float xb) {
    return __fsub_rn(__fsub_rn(__fsub_rn(u, c0), __fmul_rn(c1, xa)), __fmul_rn(c2, xb));
}

// the layer whose input gradient's conditioner path the NEXT backward layer's pass 1 finishes (instead of a tbwd3f launch)
struct PrevLayer {
    const double *tot;           // its pass-2 totals
    const float *tcanon_l, *stats_l, *ubuf;
    const float *ubuf2;          // second plane of u_k (branch-split pass 2: ubuf = logvar branch, ubuf2 = mu branch), or NULL
    const float *x;              // its input points = the output of the layer at hand
    float *dcanon_l;
    double count;
    int ka, kb, has;
    int fused_launches;          // host side: pass-1 launches of this call that carried the layer above's column sums so far
    // (this struct is a kernel argument of pass 1, which runs at its register cap: its size and layout are part of that kernel's
    // code generation -- r05: one more int here, 8 bytes with padding, cost pass 1 0.8 us; host-only state goes to BackwardCall)
};

// r04: the column sums of the layer ABOVE's pass-2 partials (the former tcolsum launch between two backward layers) ride in
// this launch as extra workgroups -- blockIdx.y < cs.rows, dispatched first -- so that the launch boundary
// tbwd2 -> tcolsum -> tbwd1 becomes tbwd2 -> tbwd1.  The 24 of them that own the 384 totals pass 1 needs (sum dh0a and the
// two centred input sums per branch and feature: P2_S, P2_A, P2_B) come first, publish their totals with agent-scope stores and
// raise a counter; the ordinary workgroups recompute the layer's pre-activations first (that needs nothing from above),
// then wait for the counter (one lane polls; see colsum_wait for the ordering), read the totals past
// their L1 / L2 (agent scope) and go on.  The other 256 column workgroups (the dW1 totals) are waited for by nobody.  If the counter does not arrive (HIP promises no dispatch
// order; observed: ascending), a workgroup gives up polling and sums the 512 columns itself, in the same order.
struct ColsumJob {
    const float *part2;          // (nrows, 2 * P2_J) pass-2 partial rows of the layer above
    double *tot;                 // 2 * P2_J totals
    float *dcanon_prev;          // the layer above's gradient block (receives dW1)
    unsigned *flag;              // arrivals of the critical column workgroups, monotonic over the call
    unsigned target;             // value of *flag once this launch's 16 have arrived
    int nrows, rows;             // partial rows; rows of the grid taken by the column workgroups (0 = none in this launch)
};
// column workgroups that own the totals pass 1 waits for (P2_S, P2_A, P2_B of both branches: 384 columns): 24 of them with 16
// columns each -- a thread then has one round of eight row loads instead of two (r05: pass 1 waited ~1 us for them)
constexpr int CS_CRIT = 24;
// first column of column workgroup `id`: the critical ones (16 columns) first, then the others (32 columns)
__device__ __forceinline__ int cs_first_column(int id) {
    constexpr int PB = P2_J / 32;                      // 134 blocks of 32 per branch: 2 (P2_S) + 128 (dW1) + 4 (P2_A, P2_B)
    if (id < CS_CRIT) { const int b = id / 12, k = id - 12 * b, k2 = k >> 1; return (b * PB + (k2 < 2 ? k2 : PB - 6 + k2)) * 32 + 16 * (k & 1); }
    int r = id - CS_CRIT;                              // the others in ascending order, skipping those
    const int b = r >= PB - 6, q = r - b * (PB - 6);
    return (b * PB + 2 + q) * 32;
}
// one column of the partials, summed exactly as tcolsum_kernel does: 32 row groups (row r in group r & 31, ascending), then
// the groups in order
__device__ __forceinline__ double colsum_group(const ColsumJob &cs, int rg, int j) {      // row group rg of column j
    const int J = 2 * P2_J;
    double s = 0;
    int r = rg;
    for (; r + 224 < cs.nrows; r += 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = cs.part2[(size_t)(r + 32 * k) * J + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < cs.nrows; r += 32) s += cs.part2[(size_t)r * J + j];
    return s;
}
__device__ __forceinline__ void colsum_role(const ColsumJob &cs, int id, uint8_t *smem) {
    double (*acc)[33] = (double (*)[33])smem;
    const bool crit = id < CS_CRIT;                                            // (workgroup-uniform)
    // 512 threads: 32 columns x row groups rg0 and rg0 + 16, or (critical) 16 columns x all 32 row groups
    const int c = crit ? threadIdx.x & 15 : threadIdx.x & 31, rg0 = crit ? threadIdx.x >> 4 : threadIdx.x >> 5;
    const int j = cs_first_column(id) + c;
    if (crit) {
        acc[rg0][c] = colsum_group(cs, rg0, j);
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) acc[rg0 + 16 * half][c] = colsum_group(cs, rg0 + 16 * half, j);
    }
    __syncthreads();
    if (rg0 == 0) {
        double t = 0;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += acc[r][c];
        __hip_atomic_store(&cs.tot[j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int b2 = j / P2_J, jj = j - b2 * P2_J;
        if (jj >= P2_W && jj < P2_W + 4096) cs.dcanon_prev[b2 * T_BR + T_W1 + jj - P2_W] = (float)t;
    }
    if (crit) {
        // (the threads that stored are the first 16: one wave -- its own vmcnt(0) covers them all, no barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the totals have left this CU before the counter moves
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cs.flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ordinary workgroups: wait until the critical totals are there (or produce them)
// (the poll is bounded: 256 round trips to the memory side, ~0.2 ms -- a column workgroup takes ~3 us; the fallbacks taken are
// counted process-wide, dpf_train_colsum_fallbacks(): tests/test_gpu_flow_train.py checks the `dispatched first` assumption)
__device__ unsigned g_colsum_fallbacks = 0;
__device__ __forceinline__ void colsum_wait(const ColsumJob &cs, int *lds_word) {
    if (threadIdx.x == 0) {
        int ok = 0;
        // Relaxed polls and NO acquire fence behind them.  r05 measured both (same box, tools/train_ab_prof.sh): an acquire on the
        // poll invalidates this CU's caches on every round trip, and ONE agent-scope acquire fence after the successful poll
        // (buffer_inv sc1 + vmcnt(0)) still costs 1.6 us per layer (tbwd1 15.4 -> 17.0 us) -- it throws away the L2 lines of the
        // weights the workgroup is about to read.  What orders the hand-off instead: the producer's totals are write-through
        // stores that have left its CU (vmcnt(0)) before its RELEASE increment; the consumer issues its loads of the totals only
        // after it has SEEN the counter (program order on one in-order memory pipe) and those loads are agent-scope atomics
        // themselves -- they are served past the L1 and the XCD's L2, where the stores are.
        for (int it = 0; it < 256; ++it) {
            if (__hip_atomic_load(cs.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cs.target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __hip_atomic_fetch_add(&g_colsum_fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_word = ok;
    }
    __syncthreads();
    if (*lds_word) return;
    // the column workgroups have not run (they were not dispatched first and there is no room beside us): their work for the
    // 384 columns pass 1 needs, by this workgroup -- one thread per column, the same 32 ordered group sums
    {
        const int J = 2 * P2_J, i = threadIdx.x;
        if (i < 384) {
            const int b = i / 192, q = i - 192 * b;
            const int j = b * P2_J + (q < 64 ? P2_S + q : P2_A - 64 + q);
            double t = 0;
            for (int k = 0; k < 32; ++k) {         // (a rolled loop with one accumulator: this path must not cost the kernel registers)
                double gk = 0;
                for (int r = k; r < cs.nrows; r += 32) gk += cs.part2[(size_t)r * J + j];
                t += gk;
            }
            __hip_atomic_store(&cs.tot[j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the same bits the column workgroup writes
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Pass 1: recompute the layer to h2, differentiate the coupling transform and the output SharedDot.
//   stores  dout (B,4,N) = d(o_logvar a,b), d(o_mu a,b)         dp_in <- direct term  g * d(p_out)/d(p)
//   partial sums per workgroup: part1[blk][br][k][64], k = 0 dW2a, 1 dW2b, 2 da, 3 dc;  part1[blk][512 + br*2 + w] = db2
// the gradient w.r.t. p_out is g_p + g_p2 (either may be NULL = zero, like g_mu and g_lv)
// (min 4 waves per SIMD = two workgroups per CU: the column workgroups of ColsumJob must find room BESIDE the ordinary ones;
// the branch-split form serves B * N / 256 <= 64 only -- 128 ordinary + 24 critical column workgroups have a CU each -- and
// is one register short at 128)
template <int NS, bool F16 = false, bool BSPLIT = false>
__global__ __launch_bounds__(TW * 64, BSPLIT ? 2 : 4) void tbwd1_kernel(TArgs a, const float *__restrict__ g_p, const float *__restrict__ g_p2,
                                                           const float *__restrict__ g_mu, const float *__restrict__ g_lv,
                                                           const float *__restrict__ mu_l, const float *__restrict__ lv_l,
                                                           float *__restrict__ dp_in, float *__restrict__ dout,
                                                           float *__restrict__ part1, PrevLayer pv, unsigned *__restrict__ tickets,
                                                           float *__restrict__ pc, float *__restrict__ dfm_l, ColsumJob cs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int L_FILM = pt_a0n(NS), L_FILMB = L_FILM + 2048, L_RED = L_FILMB + 2048;
    // grid rows [0, cs.rows): the 24 critical column workgroups (dispatched first); rows [cs.rows, cs.rows + B): the clouds;
    // rows behind them: the other column workgroups (dW1 totals: nobody in this launch waits for them)
    // BSPLIT (gridDim.z == 2), small batches: one conditioner branch per workgroup (blockIdx.z), as tbwd2_kernel
    const int br_lo = BSPLIT ? (int)blockIdx.z : 0, br_hi = BSPLIT ? (int)blockIdx.z + 1 : 2;
    if (cs.rows > 0 && ((int)blockIdx.y < cs.rows || (int)blockIdx.y >= cs.rows + a.B)) {
        if (blockIdx.z != 0) return;
        const int id = (int)blockIdx.y < cs.rows ? (int)(blockIdx.y * gridDim.x + blockIdx.x)
                                                 : CS_CRIT + (int)((blockIdx.y - cs.rows - a.B) * gridDim.x + blockIdx.x);
        const bool crit_row = (int)blockIdx.y < cs.rows;
        if ((crit_row && id < CS_CRIT) || (!crit_row && id < 2 * P2_J / 32 + CS_CRIT / 2)) colsum_role(cs, id, smem);
        return;
    }
    float *red = (float *)(smem + L_RED);                                  // per wave [2 br][4][64] + [2][2] (+4 pad)
    const int bi = (int)blockIdx.y - cs.rows, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KP(1, 0)
    // (unconditional: the host hands valid pointers even when there is no layer above -- `cl = {}` merged with the loaded
    // values cost register copies, i.e. a wait for the loads right here)
    // (with the column sums in this launch -- cs.rows > 0 -- the totals do not exist yet: loaded behind the wait below)
    CoefLoads cl = {};
    if (cs.rows == 0) cl = bwd3_loads(pv.kb >= 0 ? 2 : 1, pv.tot, pv.tcanon_l, pv.stats_l);      // first: results return in order
    asm volatile("" ::: "memory");
    // every load of the prologue is requested here, in the order its consumer comes: the pass-3 totals (above), this thread's
    // point (inputs, the layer's stored outputs, the gradients that reach it), then the weights -- one in-order stream
    const float w2_v = threadIdx.x < 256 ? a.tcanon_l[(threadIdx.x >> 7) * T_BR + T_W2 + (threadIdx.x & 127)] : 0.f;
    // f16x3: 2^-kw of each branch's packed W1 (the recomputed pre-activations below are 2^kw (h1 + D))
    const float wsinv0 = F16 ? *(const float *)(a.packed_l + pt_tail(NS)) : 1.0f, wsinv1 = F16 ? *(const float *)(a.packed_l + pt_tail(NS) + 4) : 1.0f;
    const int N = a.N, n = (blockIdx.x * TW + wave) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const size_t cloud = (size_t)bi * 3 * N;
    float p[3], gp[3], gm[3], gl[3], mus[3], lvs[3], gp1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = cloud + (size_t)c * N + nc;
        p[c] = a.p_in[o];
        mus[c] = mu_l[o];                                                   // the forward pass's own outputs of this layer
        lvs[c] = lv_l[o];
        gp[c] = g_p2 ? g_p2[o] : 0.f;                                       // what the layer above passes down (direct term so far)
        gm[c] = valid && g_mu ? g_mu[o] : 0.f;
        gl[c] = valid && g_lv ? g_lv[o] : 0.f;
    }
    const bool fused = cs.rows > 0;       // (kernel argument: uniform)
    // what only finish_gp reads: the other gradient term and the layer above's point values.  With the column sums in this
    // launch finish_gp runs BEHIND the first branch's recomputation: requested there (the wait for the totals covers the
    // round trip), so that seven registers are not carried -- spilled, at the 128 this kernel may use -- across it
    float xa2 = 0.f, xb2 = 0.f, u2a = 0.f, u2b = 0.f;
    auto load_prev = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) gp1[c] = g_p ? g_p[cloud + (size_t)c * N + nc] : 0.f;
        if (pv.has) {
            const float *xc = pv.x + cloud;
            xa2 = xc[(size_t)pv.ka * N + nc]; xb2 = pv.kb >= 0 ? xc[(size_t)pv.kb * N + nc] : 0.f;
            u2a = pv.ubuf[((size_t)bi * 2 + 0) * N + nc]; u2b = pv.kb >= 0 ? pv.ubuf[((size_t)bi * 2 + 1) * N + nc] : 0.f;
            if (pv.ubuf2 != nullptr) {     // (wave-uniform) the mu branch's share, added once: the same rounding in tbwd3f_kernel
                u2a = __fadd_rn(u2a, pv.ubuf2[((size_t)bi * 2 + 0) * N + nc]);
                if (pv.kb >= 0) u2b = __fadd_rn(u2b, pv.ubuf2[((size_t)bi * 2 + 1) * N + nc]);
            }
        }
    };
    if (!fused) load_prev();
    const StageRegs<pt_a0n(NS)> wregs = stage_load<pt_a0n(NS)>(a.packed_l, wave, lane);
    const StageRegs<2048> fregs = stage_load<2048>((const uint8_t *)(a.film_l + (size_t)bi * 512), wave, lane);
    const StageRegs<2048> fbregs = stage_load<2048>((const uint8_t *)(a.filmb_l + (size_t)bi * FB_CLOUD), wave, lane);
    // pin the consumers of the pass-3 totals BEHIND the issue of everything else (the compiler had started on them right
    // behind their loads: a cold round trip before the point and weight loads were even requested)
    asm volatile("" : "+v"(cl.S), "+v"(cl.Sa), "+v"(cl.Sb));
    asm volatile("" : "+v"(cl.ea), "+v"(cl.eb), "+v"(cl.caa), "+v"(cl.cbb), "+v"(cl.cab));
    asm volatile("" : "+v"(cl.wa), "+v"(cl.wb), "+v"(cl.gamma), "+v"(cl.rstd0));
    float *w2s = red + TW * 520;                                                // [2 br][2][64] raw sd2.weight
    if (threadIdx.x < 256) w2s[threadIdx.x] = w2_v;
    float *pcoef = w2s + 256 + TW * 64;                                         // 8 floats behind the per-wave scratch
    const int wg_lin = bi * gridDim.x + blockIdx.x;
    auto finish_gp = [&](const CoefLoads &c3) {
        if (pv.has)   // the previous backward layer's pass 3 (scratch: the reduction slots, free until the end)
            bwd3_coefs(c3, wg_lin + (int)blockIdx.z, pv.kb >= 0 ? 2 : 1, pv.count, pv.dcanon_l, (double (*)[8])red, pcoef);   // (its writer: workgroup 0 of z = 0)
        if (pv.has) {   // + the conditioner path of the layer above: dx_k = u_k - C_k - alpha_k x_a - beta_k x_b on ITS kept channels
            // every operation rounded on its own, in tbwd3f_kernel's order (the two launch forms give the same bits)
            const float xa2c = __fsub_rn(xa2, pcoef[3]), xb2c = __fsub_rn(xb2, pcoef[7]);
            const float ta = cond_path(u2a, pcoef[0], pcoef[1], pcoef[2], xa2c, xb2c);
            const float tb = pv.kb >= 0 ? cond_path(u2b, pcoef[4], pcoef[5], pcoef[6], xa2c, xb2c) : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c == pv.ka || c == pv.kb) gp[c] = __fadd_rn(gp[c], c == pv.ka ? ta : tb);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gp[c] = valid ? __fadd_rn(gp1[c], gp[c]) : 0.f;
    };
    if (!fused) finish_gp(cl);            // the totals were there at launch: while the weights land
    stage_store(wregs, smem + L_PACK, wave, lane);
    stage_store(fregs, smem + L_FILM, wave, lane);
    stage_store(fbregs, smem + L_FILMB, wave, lane);
    const float xa = sel3(a.ka, p[0], p[1], p[2]), xb = a.kb >= 0 ? sel3(a.kb, p[0], p[1], p[2]) : 0.f;
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    KP(1, 1)
    __syncthreads();
    KP(1, 2)
    const float *film = (const float *)(smem + L_FILM);
    const float *filmb = (const float *)(smem + L_FILMB);
    // the pre-activations h1 + D of one branch in the SWAPPED orientation (see below); needs nothing from the layer above
    auto recompute_pre = [&](int br, f32x16 (&pre)[2]) {
        f32x16 acc0[2];
        u32x4 bf[NS][4];
        input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, acc0);
        split_fragment<true, NS, false, F16>(acc0, bf, a.negone);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float dsh = film[br * FILM_BR_FLOATS + 32 * t + pl];
#pragma unroll
            for (int r = 0; r < 16; ++r) pre[t][r] = dsh;
        }
        chain_mfma_swapped<NS, F16>(smem + L_PACK + PT_A1, br, lane, bf, pre);
    };
    f32x16 pre0[2];
    if (fused) {
        // the first branch's recomputation first: by the time it is done the critical column workgroups of this launch have
        // long published their totals (they read 512 columns of the partials; this is ~1 400 instructions)
        recompute_pre(br_lo, pre0);
        KP(7, 0)
        load_prev();
        CoefLoads c3;
        bwd3_loads_const(c3, pv.kb >= 0 ? 2 : 1, pv.tcanon_l, pv.stats_l);    // (requested in front of the wait: only the three totals come from the column sums)
        colsum_wait(cs, (int *)(pcoef + 12));
        KP(7, 1)
        bwd3_loads_totals<true>(c3, pv.tot);
        finish_gp(c3);
        KP(7, 2)
    }
    // ---- coupling transform and its derivative (flows.py:96-115)
    const bool inverse = a.mode == DPF_MODE_INVERSE;
    float dmu_w[2] = {0.f, 0.f}, dlv_w[2] = {0.f, 0.f};
#pragma unroll
    // mu and logvar = softsign(o_logvar) of the warped channels are the forward kernel's stored outputs (r02: no third
    // recomputation of the conditioner here); d softsign / d o = 1 / (1 + |o|)^2 = (1 - |logvar|)^2
    for (int c = 0; c < 3; ++c) {
        const bool isa = c == a.wa, isb = c == a.wb;
        const float lv = (isa || isb) ? lvs[c] : 0.f;
        const float mu = (isa || isb) ? mus[c] : 0.f;
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
        if (valid && h == 0 && blockIdx.z == 0) dp_in[cloud + (size_t)c * N + n] = dpc;       // direct term; pass 3 adds the conditioner path
        const float dsoft = (1.0f - fabsf(lv)) * (1.0f - fabsf(lv));
        if (isa) { dmu_w[0] = dmu; dlv_w[0] = dlv * dsoft; }
        if (isb) { dmu_w[1] = dmu; dlv_w[1] = dlv * dsoft; }
    }
    if (valid && blockIdx.z == 0) {                                        // half 0 stores the logvar pair, half 1 the mu pair
        float *d = dout + ((size_t)bi * 4 + 2 * h) * N + n;
        d[0] = h ? dmu_w[0] : dlv_w[0];
        d[N] = h ? dmu_w[1] : dlv_w[1];
    }
    // ---- output SharedDot backward, FiLM backward, per-feature sums: the pre-activations once more, in the SWAPPED
    // orientation (lane = feature 32 t + pl, register r = point (r & 3) + 8 (r >> 2) + 4 h; same products in the same order, so
    // the same bits) -- the per-feature constants are per-lane and the sums over the tile's points are in-lane adds plus one
    // cross-half add, where the lane = point layout needed four 31-shuffle butterflies per branch
    float *pts = w2s + 256 + wave * 64;                                    // per-wave scratch: d(o) of the tile's 32 points [2][32]
    auto branch_sums = [&](const int br) {
        const float doa = br == 0 ? dlv_w[0] : dmu_w[0], dob = br == 0 ? dlv_w[1] : dmu_w[1];
        if (!h) { pts[pl] = doa; pts[32 + pl] = dob; }
        f32x16 pre[2];
        if (fused && br == br_lo) { pre[0] = pre0[0]; pre[1] = pre0[1]; }
        else recompute_pre(br, pre);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 doa4[4], dob4[4];
        // opaque to the optimiser: the offsets stay "lane base + immediate" (as known bits they become OR-ed constants, one
        // live register each); made here from the thread index, not carried through the recomputation above
        int h4 = 4 * (int)((threadIdx.x >> 5) & 1);
        asm volatile("" : "+v"(h4));
#pragma unroll
        for (int q = 0; q < 4; ++q) { doa4[q] = *(const f32x4 *)(pts + 8 * q + h4); dob4[q] = *(const f32x4 *)(pts + 32 + 8 * q + h4); }
        float *rw = red + wave * 520;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int fo = 32 * t + pl;
            const float fa = filmb[br * FB_BR + 0 * 64 + fo] * filmb[br * FB_BR + 2 * 64 + fo];      // h2 = FA * relu(pa)
            const float rstd1 = filmb[br * FB_BR + 2 * 64 + fo], ca = filmb[br * FB_BR + 3 * 64 + fo];
            const float w2a = w2s[br * 128 + fo], w2b = w2s[br * 128 + 64 + fo];
            // r03: 8 VALU per element instead of 11 (the kernel is bound by VALU issue): the per-lane factor FA leaves the loop,
            // and sum dh2a * pa = sum lin * relu(pa) = W2a * sum do_a relu(pa) + W2b * sum do_b relu(pa) needs no loop at all
            float s0 = 0.f, s1 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float da = doa4[r >> 2][r & 3], db = dob4[r >> 2][r & 3];
                const float rp = relu(pre[t][r]);
                s0 = __builtin_fmaf(da, rp, s0);                                               // dW2[w] = FA * sum do_w * relu(pa)
                s1 = __builtin_fmaf(db, rp, s1);
                s3 += dh2a_of(pre[t][r], w2a, w2b, da, db);                                    // dc = sum dh2a,  dh2a = [pa > 0] sum_w W2[w]*do_w
            }
            if constexpr (F16) { const float u = br ? wsinv1 : wsinv0; s0 *= u; s1 *= u; }     // sums of do * relu(2^kw pa): back to pa's scale (exact)
            float s2 = rstd1 * (w2a * s0 + w2b * s1) - ca * s3;                                // da = sum dh2a * h1n,  h1n = pa * rstd1 - c/a
            s0 *= fa; s1 *= fa;
            s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
            if (!h) {
                rw[(br * 4 + 0) * 64 + fo] = s0;
                rw[(br * 4 + 1) * 64 + fo] = s1;
                rw[(br * 4 + 2) * 64 + fo] = s2;
                rw[(br * 4 + 3) * 64 + fo] = s3;
            }
        }
        float sa = h == 0 ? doa : 0.f, sb = h == 0 ? dob : 0.f;                   // db2: each point once
        for (int q = 32; q > 0; q >>= 1) { sa += __shfl_xor(sa, q); sb += __shfl_xor(sb, q); }
        if (lane == 0) { rw[512 + br * 2 + 0] = sa; rw[512 + br * 2 + 1] = sb; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                   // the scratch is rewritten for the next branch
    };
    if constexpr (BSPLIT) {
        branch_sums((int)blockIdx.z);
    } else {
        KP(7, 3)
        branch_sums(0);
        KP(7, 4)
        branch_sums(1);
    }
    KP(1, 3)
    __syncthreads();
    const size_t blk = (size_t)wg_lin;
    for (int i = threadIdx.x; i < 516; i += TW * 64) {
        const int ibr = i < 512 ? i >> 8 : (i - 512) >> 1;                 // the branch entry i belongs to
        if (ibr < br_lo || ibr >= br_hi) continue;                         // (branch split: the other workgroup's half of the row)
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < TW; ++w) t += red[w * 520 + i];
        // written through (agent scope): with the ticket below, the last workgroup of this cloud reads it, possibly from another XCD
        __hip_atomic_store(&part1[blk * 520 + i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    KP(1, 4)
    // Who finishes pass 1 (the per-cloud totals of these rows, the FiLM gradients, and -- for pass 2 -- the BN1-backward means)?
    // r05 measured both forms (same box, tools/train_ab_prof.sh).  tickets == nullptr: 17 role workgroups at the
    // front of pass 2's launch do it once (MeansJob) and this kernel ends with its row -- 2.0 us shorter; pass 2 needs one CU per
    // workgroup, so the role workgroups pay only where CUs are idle (small batches: <= 239 workgroups).  Otherwise, as r02-r04:
    if (tickets == nullptr) return;
    // ---- per-cloud totals of pass 1 and the FiLM gradients of the cloud (formerly the tcloudsum launch), by whichever of the
    // cloud's workgroups arrives last: partial rows published with write-through stores, a ticket per cloud, rows read back
    // with agent-scope loads (per-XCD L2s are not coherent) and added in workgroup order -- the same sums in the same order
    // wherever the workgroups ran.  The ticket returns to zero for the next layer.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this workgroup's row has left the CU
    __syncthreads();
    unsigned *tk = (unsigned *)(w2s + 256 + TW * 64 + 8);                  // one word behind the pass-3 coefficients
    if (threadIdx.x == 0)
        *tk = __hip_atomic_fetch_add(&tickets[bi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int nb = gridDim.x;
    if (*tk != (unsigned)(nb * (int)gridDim.z - 1)) return;               // (branch split: two arrivals per row)
    for (int j = threadIdx.x; j < 516; j += TW * 64) {
        float s = 0.f;
        for (int k0 = 0; k0 < nb; k0 += 8) {                               // eight loads in flight, added in order
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = k0 + i < nb ? __hip_atomic_load(&part1[((size_t)bi * nb + k0 + i) * 520 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        pc[(size_t)bi * 520 + j] = s;
        if (j < 512) {
            const int br = j >> 8, k = (j >> 6) & 3, f = j & 63;
            if (k == 2) dfm_l[((size_t)(br * 2 + 0) * a.B + bi) * 64 + f] = s * (filmb[br * FB_BR + f] - a.eps);
            if (k == 3) dfm_l[((size_t)(br * 2 + 1) * a.B + bi) * 64 + f] = s;
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(&tickets[bi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (the BN1-backward means and dW2 / db2 -- formerly a one-workgroup tfinish1 kernel -- are computed in tbwd2's prologue)

// Pass 2: BN1 backward, dh0 = W1^T dh1 (matrix cores), dW1 = dh1 h0^T (matrix cores, contraction over the tile's 32 points: both
// operands are K = points fragments straight from swapped-orientation accumulators), relu backward.
//   part2[blk][br][P2_J]: P2_S sum dh0a (d beta0), P2_W dW1 (row = out feature), P2_A / P2_B sum dh0a * (x_k - E x_k)
//   ubuf (2, B, 2, N): plane br = u_k of branch br's features, u_k = sum_f W0[f][k]*rstd0*gamma0 * dh0a[f]; the consumer adds the planes
//
// r05 -- a wave owns ONE conditioner branch.  Until r04 a wave ran both branches of one 32-point tile one after the other and
// every branch ended in a workgroup-wide reduction of its 64 x 64 dW1 tile: 12 barriers per kernel.  The stamps
// (tools/train_phase_prof.py, profiles/r04) showed what they cost: of the two waves of a SIMD the older one wins the issue
// arbitration and gets through a branch in 11.6 K ticks, its partner in 14.3 K, and the older then WAITS 6 K ticks at the
// reduction's barriers -- twice per kernel -- only to start the next branch phase-aligned with its partner again (matrix
// phases beside matrix phases, which serialise on the SIMD's one matrix pipe).  PAIR: waves 0-3 run the logvar branch of
// tiles w and w + 4 of the workgroup's eight, waves 4-7 the mu branch of the same tiles.  dW1 of a branch accumulates over the
// wave's two tiles in the MFMA accumulators themselves, there is no barrier between the prologue and the one reduction at
// the end (four waves per branch instead of eight: half the LDS traffic, 5 barriers instead of 12), and the skew the
// arbitration creates is kept: a wave's VALU stretches run beside its partner's MFMA chains for the whole kernel.
// !PAIR (small batches: at most half a workgroup per CU, i.e. B * N / 256 <= 128): one branch per WORKGROUP (blockIdx.z), a
// tile per wave, as r04's SPLIT.
// The instruction diet that came with it (the kernel is bound by vector issue): the BN1-backward formula's point-dependent
// part K1[f] do_a[pt] + K2[f] do_b[pt] + C0[f] is ONE input-style MFMA per feature tile (3-way bf16 splits in the 16 K slots:
// fp32-accurate, as the layer's input contraction) instead of 3 VALU per element; sum dh0a * h0n is derived from the two
// centred input sums (bwd3_coefs) -- one input MFMA pair and an FMA per element gone; relu(h0) arrives scaled by 16 from an
// input fragment scaled by 16 (exact) -- no multiply per element in front of its fp16 split; dW1 is unscaled by the reducer
// (an FMA where it had an add) instead of 64 multiplies per wave and tile.
// r05: what stands between pass 1 and pass 2 -- the per-cloud totals of pass 1's rows, the FiLM gradients, dW2 / db2 and the
// BN1-backward means over the batch -- is done ONCE, by 17 role workgroups dispatched at the front of pass 2's grid (32 of
// the row's 516 columns each), instead of r02-r04's per-cloud ticket at the end of pass 1 (write-through row, vmcnt(0), atomic
// round trip, the last arriver reads the cloud's rows back: 2.6 us of every pass 1) plus a recomputation of the means by
// every workgroup of pass 2.  The ordinary workgroups stage their 70 KB of weights first -- that needs nothing from pass 1 --
// and then wait for the role workgroups' counter (they are ~3 us of loads and sums; the staging takes ~5); the 256 means
// are read past the L1 / L2 (agent scope).  Same pattern, same ordering argument and the same bounded poll as the column
// sums inside pass 1 (colsum_wait); a workgroup whose poll runs out sums the 256 columns it needs itself, in the role
// workgroups' order (all workgroups of a launch must hold the same bits).
struct MeansJob {
    const float *part1;          // (B * nb, 520) rows of pass 1
    const float *pc;             // (B, 520) the per-cloud totals, when pass 1 finished them itself (no role workgroups: rows == 0)
    float *dfm_l;                // FiLM gradients of the layer [br][sub][B][64]
    float *s12;                  // [br][2][64]: mean dh1n, mean dh1n * h1n -- published by the role workgroups
    unsigned *flag;              // arrivals of the role workgroups, monotonic over the call
    unsigned target;             // value of *flag once this launch's have arrived
    int nb;                      // rows (workgroups of pass 1) per cloud
    int rows;                    // rows of the grid taken by the role workgroups
};
constexpr int MJ_ROLES = (516 + 31) / 32;
// total of column j of cloud b's rows, in row order, eight loads in flight
__device__ __forceinline__ float cloud_col_total(const float *__restrict__ part1, int nb, int b, int j) {
    float s = 0.f;
    for (int k0 = 0; k0 < nb; k0 += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = k0 + i < nb ? part1[((size_t)b * nb + k0 + i) * 520 + j] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    return s;
}
__device__ __forceinline__ void means_role(const TArgs &a, const MeansJob &mj, int id, double count, float *__restrict__ dcanon_l, uint8_t *smem) {
    double (*acc)[33] = (double (*)[33])smem;
    KPR(0)
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5, j = id * 32 + c;       // 512 threads: 32 columns x 16 cloud groups
    const bool col = j < 512;
    const int brj = col ? j >> 8 : (j - 512) >> 1, k = col ? (j >> 6) & 3 : 4, f = j & 63;
    double s = 0;
    if (j < 516)
        for (int b0 = rg; b0 < a.B; b0 += 32) {       // two clouds per round: their rows' loads are all in flight together
            const int b1 = b0 + 16;
            const bool two = b1 < a.B;
            const float av0 = k == 2 || k == 3 ? a.filmb_l[(size_t)b0 * FB_CLOUD + brj * FB_BR + f] : 1.0f;
            const float av1 = two && (k == 2 || k == 3) ? a.filmb_l[(size_t)b1 * FB_CLOUD + brj * FB_BR + f] : 1.0f;
            const float pc0 = cloud_col_total(mj.part1, mj.nb, b0, j);
            const float pc1 = two ? cloud_col_total(mj.part1, mj.nb, b1, j) : 0.f;
            if (k == 2 || k == 3) {       // da, dc of the cloud's FiLM vectors; their a-weighted batch sums are the BN1-backward means
                mj.dfm_l[((size_t)(brj * 2 + (k == 2 ? 0 : 1)) * a.B + b0) * 64 + f] = k == 2 ? pc0 * (av0 - a.eps) : pc0;
                if (two) mj.dfm_l[((size_t)(brj * 2 + (k == 2 ? 0 : 1)) * a.B + b1) * 64 + f] = k == 2 ? pc1 * (av1 - a.eps) : pc1;
            }
            s += (double)av0 * pc0;       // (dW2 / db2: plain batch totals, av = 1)
            if (two) s += (double)av1 * pc1;
        }
    KPR(1)
    acc[rg][c] = s;
    __syncthreads();
    KPR(2)
    if (rg == 0 && j < 516) {
        double t = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[r][c];
        if (k == 3) __hip_atomic_store(&mj.s12[(brj * 2 + 0) * 64 + f], (float)(t / count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // dh1n = a * dh2a
        else if (k == 2) __hip_atomic_store(&mj.s12[(brj * 2 + 1) * 64 + f], (float)(t / count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // dh1n * h1n
        else if (k < 2) dcanon_l[brj * T_BR + T_W2 + k * 64 + f] = (float)t;
        else { dcanon_l[brj * T_BR + T_B2 + (j & 1)] = (float)t; dcanon_l[brj * T_BR + T_B2 + 2 + (j & 1)] = 0.f; }
    }
    // (the threads that stored are the first 32: one wave -- its own vmcnt(0) covers them all, no barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the means have left this CU before the counter moves
    KPR(3)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(mj.flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    KPR(4)
}
// ordinary workgroups: the 256 means into LDS (s12s), from the role workgroups or -- if they have not run -- by this workgroup
__device__ unsigned g_means_fallbacks = 0;
__device__ __forceinline__ void means_wait(const TArgs &a, const MeansJob &mj, double count, float *s12s, int *lds_word) {
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < 256; ++it) {       // (relaxed polls, no fence: see colsum_wait)
            if (__hip_atomic_load(mj.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= mj.target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) __hip_atomic_fetch_add(&g_means_fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_word = ok;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int i = threadIdx.x;
        float v;
        if (*lds_word) {
            v = __hip_atomic_load(&mj.s12[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {      // the role workgroups' sums for this thread's column, in their order: 16 cloud groups, clouds ascending within a group
            const int brj = i >> 7, k = (i >> 6) & 1 ? 2 : 3, f = i & 63, j = brj * 256 + k * 64 + f;
            double t = 0;
            for (int rg = 0; rg < 16; ++rg) {
                double sg = 0;
                for (int b = rg; b < a.B; b += 16)
                    sg += (double)a.filmb_l[(size_t)b * FB_CLOUD + brj * FB_BR + f] * cloud_col_total(mj.part1, mj.nb, b, j);
                t += sg;
            }
            v = (float)(t / count);
        }
        s12s[i] = v;
    }
}

#ifdef DPF_PROFILE
__device__ unsigned long long *g_tprof = nullptr;
#define TP(i) { __builtin_amdgcn_sched_barrier(0); if (ti == 0) tt[i] = __builtin_amdgcn_s_memtime(); else if ((i) == 0 || (i) == 6) tt[(i) == 0 ? 7 : 8] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
// (the phase boundaries fence the instruction scheduler in every build: left free it starts a phase's loads and conversions
// inside the previous one, and the kernel -- at the 256 registers two waves per SIMD may have -- spills)
#define TP(i) __builtin_amdgcn_sched_barrier(0);
#endif
// x of lane (i ^ 32) + x of lane i, on the VALU (v_permlane32_swap; __shfl_xor is an LDS round trip)
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    return u2f(r[0]) + u2f(r[1]);
}
// the same for the two 16-lane rows of each wave half
__device__ __forceinline__ float row_pair_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(f2u(x), f2u(x), false, false);
    return u2f(r[0]) + u2f(r[1]);
}
// x as seen through a DPP lane pattern (the compiler folds the move into the consuming VALU instruction)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) { return u2f(__builtin_amdgcn_update_dpp(0u, f2u(x), CTRL, 0xF, 0xF, true)); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141;
// Sums over the 32 lanes of a wave half for 16 registers at once (swapped orientation: lanes = features, registers = the
// half's 16 points): recursive halving -- a lane keeps one half of its registers, its partner the other, each adds what the
// other sends; 15 exchanges instead of 16 x 5, every one a DPP operand of the add (no LDS, no v_mov).  The partners: 15 - i
// within a row (row_mirror: flips lane bit 3), 7 - i within a half row (flips bit 2, keeps bit 3: the two hold the same
// registers), i ^ 2, i ^ 1, then the other row of the half.  On return w[0] of lane i is the total of register i & 15.
template <int CTRL, int NH>
__device__ __forceinline__ void halve_stage(float (&w)[16], bool up) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const float keep = up ? w[i + NH] : w[i], send = up ? w[i] : w[i + NH];
        w[i] = keep + dpp_f<CTRL>(send);
    }
}
__device__ __forceinline__ float reduce_lanes16(float (&w)[16], int lane) {
    halve_stage<DPP_ROW_MIRROR, 8>(w, (lane >> 3) & 1);
    halve_stage<DPP_ROW_HALF_MIRROR, 4>(w, (lane >> 2) & 1);
    halve_stage<DPP_XOR2, 2>(w, (lane >> 1) & 1);
    halve_stage<DPP_XOR1, 1>(w, lane & 1);
    return row_pair_sum(w[0]);
}
// the largest of a non-negative x over the wave, as its bit pattern in a scalar register: four DPP maxima within the rows
// (each row ends up uniform), then the four rows on the scalar unit (non-negative floats order like their bit patterns)
__device__ __forceinline__ uint32_t wave_max_bits(float x) {
    x = fmaxf(x, dpp_f<DPP_XOR1>(x));
    x = fmaxf(x, dpp_f<DPP_XOR2>(x));
    x = fmaxf(x, dpp_f<DPP_ROW_HALF_MIRROR>(x));
    x = fmaxf(x, dpp_f<DPP_ROW_MIRROR>(x));
    const uint32_t a = __builtin_amdgcn_readlane(f2u(x), 0), b = __builtin_amdgcn_readlane(f2u(x), 16);
    const uint32_t c = __builtin_amdgcn_readlane(f2u(x), 32), d = __builtin_amdgcn_readlane(f2u(x), 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// The operands of an input-style MFMA (flow_common.h: input_fragment / input_weight_slots8) with ROUND-TO-NEAREST parts.  The
// truncating split of the forward path gives every part the sign of the value, so the three cross terms the 16 K slots have no
// room for (mid x lo, lo x mid, lo x lo: ~2^-24 of the product) all have the product's sign -- a bias that survives a sum over
// 65 536 points.  A gradient whose terms cancel to 1e-3 of their magnitudes (the biases of the later layers) sees it: with the
// truncating parts nvp3's mu bias at (8, 2048, direct) moved from 4 x to 8 x the fp32 tensor ops' own error.  Rounded parts
// leave remainders of either sign.
__device__ __forceinline__ void split3_rne(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    hi = bf16_rne(x);
    const float r1 = x - u2f(hi << 16);
    mid = bf16_rne(r1);
    lo = bf16_rne(r1 - u2f(mid << 16));
}
__device__ __forceinline__ u32x4 input_fragment_rne(float x, int h) {
    uint32_t xh, xm, xl;
    split3_rne(x, xh, xm, xl);
    u32x4 b0;
    b0.x = xh | (xm << 16);                 // e0 = xh, e1 = xm
    b0.y = xh | (xl << 16);                 // e2 = xh, e3 = xl
    b0.z = xm | (xh << 16);                 // e4 = xm, e5 = xh
    b0.w = h ? 0x00003F80u : 0x3F803F80u;   // e6 = 1, e7 = (h == 0)
    return b0;
}
__device__ __forceinline__ u32x4 input_weight_slots8_rne(float w, float T, int h) {
    uint32_t wh, wm, wl, Th, Tm, Tl;
    split3_rne(w, wh, wm, wl);
    split3_rne(T, Th, Tm, Tl);
    u32x4 v;
    v.x = wh | (wh << 16);                      // j = 0, 1
    v.y = wm | (wh << 16);                      // j = 2, 3
    v.z = wm | (wl << 16);                      // j = 4, 5
    v.w = h == 0 ? (Th | (Tm << 16)) : Tl;      // j = 6, 7
    return v;
}
// B operand of an input-style MFMA whose point values are scaled by 16 (exact: every split part and the bias slots scale)
__device__ __forceinline__ u32x4 input_fragment_x16(float x, int h) {
    u32x4 b0 = input_fragment(x * 16.0f, h);
    b0.w = h ? 0x00004180u : 0x41804180u;   // e6 = 16, e7 = 16 (h == 0)
    return b0;
}
template <int NS, bool F16 = false, bool PAIR = true, bool ROLES = false>
__global__ __launch_bounds__(TW * 64) void tbwd2_kernel(TArgs a, MeansJob mj, double count, float *__restrict__ dcanon_l,
                                                        const float *__restrict__ dout,
                                                        float *__restrict__ ubuf, float *__restrict__ part2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef DPF_PROFILE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    KP(2, 0)
    constexpr int NT = PAIR ? 2 : 1;          // tiles per wave
    constexpr int NWB = PAIR ? 4 : 8;         // waves per branch
    constexpr int NB = PAIR ? 2 : 1;          // branches this workgroup works on
    constexpr int L_FILM = l_film(NS), L_FILMB = l_filmb(NS), L_RED = l_red(NS);
    constexpr int L_GFR = L_PACK + pt_a0n(NS);                             // dh1 input-style fragments [br][t][lane] (4 KiB, built below)
    float *uns = (float *)(smem + L_RED);                                  // per wave: what its dW1 accumulators are multiplied by at the end
    float *cf = (float *)(smem + L_RED) + 256;                             // c_fk [2 br][2][64]
    float *redw = (float *)(smem + L_RED + 4096);                          // per-wave slots (8 KB) of the workgroup reduction; their head is the wave's per-point scratch before that
    // ROLES: grid rows [0, mj.rows) are the role workgroups (dispatched first); the rows behind them: the clouds
    if (ROLES && (int)blockIdx.y < mj.rows) {
        const int id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (blockIdx.z == 0 && id < MJ_ROLES) means_role(a, mj, id, count, dcanon_l, smem);
        return;
    }
    const int bi = (int)blockIdx.y - (ROLES ? mj.rows : 0), lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int br = PAIR ? wave >> 2 : (int)blockIdx.z;                     // this wave's branch (wave-uniform)
    int h4 = 4 * h;                   // opaque to the optimiser: feature offsets stay "lane base + immediate"
    asm volatile("" : "+v"(h4));      // (as known bits they become OR-ed constants, one live register each)
    KP(3, 0)
    // r03: the loads of the prologue's tiny reductions are issued BEFORE the 70-90 KB of weight loads -- vector-memory results
    // return in order, so issued behind it they waited for all of it (tools/train_kprof.py: 5.8 K ticks for 24 loads)
    const int mq = threadIdx.x & 127, mbr = mq >> 6, mf = mq & 63, mg4 = threadIdx.x >> 7;     // means: (branch, feature) x clouds mg4, mg4 + 4, ...
    float m_av[8], m_q3[8], m_q2[8];
    float f_wa[8], f_wb[8], f_bb[8];      // workgroup (0, 0) only: the dW2 / db2 totals it writes (r03: these were 8 dependent round trips
                                          // of loads inside the means' loop -- the one workgroup every launch waited for)
    const bool first_wg = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    const float *pcs = mj.pc;             // !ROLES: the per-cloud totals pass 1 finished by ticket
    // branch-free: a cloud index past the batch is clamped and the value dropped where it is used (predicated loads became
    // exec-masked branches whose joins waited for everything in flight)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int b = mg4 + 4 * jj, bc = b < a.B ? b : 0;
        const float *qq = pcs + (size_t)bc * 520 + mbr * 256;
        m_av[jj] = ROLES ? 0.f : a.filmb_l[(size_t)bc * FB_CLOUD + mbr * FB_BR + mf];
        m_q3[jj] = ROLES ? 0.f : qq[3 * 64 + mf];
        m_q2[jj] = ROLES ? 0.f : qq[2 * 64 + mf];
    }
    if (!ROLES && first_wg) {             // (wave-uniform)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int b = mg4 + 4 * jj, bc = b < a.B ? b : 0;
            const float *qq = pcs + (size_t)bc * 520 + mbr * 256;
            f_wa[jj] = qq[0 * 64 + mf];
            f_wb[jj] = qq[1 * 64 + mf];
            f_bb[jj] = pcs[(size_t)bc * 520 + 512 + mbr * 2 + (mf & 1)];
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f_wa[jj] = f_wb[jj] = f_bb[jj] = 0.f;
    }
    // ... and so are the small table / per-point loads (registers now, LDS writes after the weight loads are issued)
    // (both halves of the workgroup load both tables, branch-free: an if / else around the loads shared a register between
    // a pending load and a constant, and the compiler put `s_waitcnt vmcnt(0)` in front of the weights' issue)
    float cf_w, cf_r, cf_g, w2_v;
    {
        const int i = threadIdx.x & 255;
        const int b2 = i >> 7, k = (i >> 6) & 1, f = i & 63;                 // c_fk = W0[f][k] * rstd0_f * gamma0_f
        const float *cb = a.tcanon_l + b2 * T_BR;
        const int nk = a.kb >= 0 ? 2 : 1;
        cf_w = cb[T_W0 + f * nk + (k < nk ? k : 0)]; cf_r = a.stats_l[b2 * ST_BR + 64 + f]; cf_g = cb[T_G0 + f];
        if (k >= nk) cf_g = 0.f;
        w2_v = a.tcanon_l[(i >> 7) * T_BR + T_W2 + (i & 127)];
    }
    // this wave's points: the kept coordinates and d(o) of ITS branch, for each of its tiles
    const int N = a.N;
    float xa_t[NT], xb_t[NT], doa_t[NT], dob_t[NT];
    {
        const float *pc = a.p_in + (size_t)bi * 3 * N;
        const float *dq = dout + ((size_t)bi * 4 + 2 * br) * N;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int tile = PAIR ? (wave & 3) + 4 * ti : wave;
            const int n = (blockIdx.x * TW + tile) * TILE + pl, nc = n < N ? n : N - 1;
            xa_t[ti] = pc[(size_t)a.ka * N + nc]; xb_t[ti] = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
            doa_t[ti] = dq[nc]; dob_t[ti] = dq[(size_t)N + nc];
        }
    }
    const float ea = a.stats_l[ST_MOM + 0], eb = a.stats_l[ST_MOM + 1];   // batch means of the kept coordinates (scalar loads)
    asm volatile("" ::: "memory");
    // the packed block without the 4 KiB nobody fills (pt_a0n): [0, pt_a0n) = W1 fragments + input fragments, [pt_a1t, end) = W1^T
    const StageRegs<pt_a0n(NS)> wregs = stage_load<pt_a0n(NS)>(a.packed_l, wave, lane);
    const StageRegs<pt_bytes(NS) - pt_a1t(NS)> wregs_t = stage_load<pt_bytes(NS) - pt_a1t(NS)>(a.packed_l + pt_a1t(NS), wave, lane);
    const StageRegs<2048> fregs = stage_load<2048>((const uint8_t *)(a.film_l + (size_t)bi * 512), wave, lane);
    const StageRegs<2048> fbregs = stage_load<2048>((const uint8_t *)(a.filmb_l + (size_t)bi * FB_CLOUD), wave, lane);
    if constexpr (!ROLES) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)    // pin the consumers of the small loads BEHIND the issue of the weight loads (else the sums are scheduled
            asm volatile("" : "+v"(m_av[jj]), "+v"(m_q3[jj]), "+v"(m_q2[jj]), "+v"(f_wa[jj]), "+v"(f_wb[jj]), "+v"(f_bb[jj]));   // right behind each load: 8 serial round trips)
    }
    KP(3, 1)
    const float *film = (const float *)(smem + L_FILM);
    const float *filmb = (const float *)(smem + L_FILMB);
    const size_t blk = (size_t)bi * gridDim.x + blockIdx.x;
    float *s12s = cf + 256;                                                // [2 br][2][64] BN1 backward means
    float *w2s = s12s + 256;                                               // [2 br][2][64] raw sd2.weight
    if (threadIdx.x < 256) cf[threadIdx.x] = cf_w * cf_r * cf_g;
    else w2s[threadIdx.x - 256] = w2_v;
    KP(3, 2)
    if constexpr (!ROLES)
    {   // BN1-backward means s12[br][2][64] = (sum dh1n, sum dh1n*h1n) / P over the per-cloud totals of pass 1 -- what
        // the one-workgroup tfinish1 kernel did, recomputed by every workgroup (same sums in the same order, under the
        // weight loads; a dependent tiny launch costs ~4.5 us); workgroup 0 also writes dW2 / db2.  Scratch: redw.
        double (*acc)[5][128] = (double (*)[5][128])redw;
        const int q = mq, br_ = mbr, f_ = mf, g4 = mg4;
        const bool first = first_wg;
        double S1 = 0, S2 = 0, w2a = 0, w2b = 0, bb = 0;
        // the first 32 clouds from the registers loaded at the top (straight-line code: inside the loop below the compiler
        // cannot tell these loads from the loop's own and waits for everything in flight, the weights included) ...
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const bool ok = g4 + 4 * jj < a.B;
            const float av = ok ? m_av[jj] : 0.f;
            S1 += (double)av * m_q3[jj];                                       // dh1n = a * dh2a
            S2 += (double)av * m_q2[jj];                                       // dh1n * h1n
            w2a += ok ? f_wa[jj] : 0.f;                                        // (zeros unless workgroup (0, 0))
            w2b += ok ? f_wb[jj] : 0.f;
            bb += ok && f_ < 2 ? f_bb[jj] : 0.f;
        }
        // ... and further rounds of 32 for bigger batches, in the same order
        for (int b0 = 32; b0 < a.B; b0 += 32) {
            float r_av[8], r_q3[8], r_q2[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int b = b0 + g4 + 4 * jj;
                const bool ok = b < a.B;
                const float *qq = pcs + (size_t)(ok ? b : 0) * 520 + br_ * 256;
                r_av[jj] = ok ? a.filmb_l[(size_t)b * FB_CLOUD + br_ * FB_BR + f_] : 0.f;
                r_q3[jj] = ok ? qq[3 * 64 + f_] : 0.f;
                r_q2[jj] = ok ? qq[2 * 64 + f_] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                S1 += (double)r_av[jj] * r_q3[jj];
                S2 += (double)r_av[jj] * r_q2[jj];
            }
            if (first)
                for (int jj = 0; jj < 8; ++jj) {
                    const int b = b0 + g4 + 4 * jj;
                    if (b >= a.B) break;
                    const float *qq = pcs + (size_t)b * 520 + br_ * 256;
                    w2a += qq[0 * 64 + f_]; w2b += qq[1 * 64 + f_];
                    if (f_ < 2) bb += pcs[(size_t)b * 520 + 512 + br_ * 2 + f_];
                }
        }
        KP(3, 3)
        acc[g4][0][q] = S1; acc[g4][1][q] = S2; acc[g4][2][q] = w2a; acc[g4][3][q] = w2b; acc[g4][4][q] = bb;
        KP(3, 4)
        lds_barrier();
        KP(3, 5)
        if (threadIdx.x < 128) {
            double S1 = 0, S2 = 0, w2a = 0, w2b = 0, bb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { S1 += acc[k][0][q]; S2 += acc[k][1][q]; w2a += acc[k][2][q]; w2b += acc[k][3][q]; bb += acc[k][4][q]; }
            s12s[(br_ * 2 + 0) * 64 + f_] = (float)(S1 / count);
            s12s[(br_ * 2 + 1) * 64 + f_] = (float)(S2 / count);
            if (first) {
                dcanon_l[br_ * T_BR + T_W2 + f_] = (float)w2a;
                dcanon_l[br_ * T_BR + T_W2 + 64 + f_] = (float)w2b;
                if (f_ < 4) dcanon_l[br_ * T_BR + T_B2 + f_] = (float)bb;
            }
        }
    }
    stage_store(wregs, smem + L_PACK, wave, lane);                        // the weights, landed while the means were formed
    stage_store(wregs_t, smem + L_PACK + pt_a1t(NS), wave, lane);
    stage_store(fregs, smem + L_FILM, wave, lane);
    stage_store(fbregs, smem + L_FILMB, wave, lane);
    static_assert(!ROLES || !PAIR, "the role form serves the one-tile-per-wave kernel");
    // ---- forward of a tile: h0 (lane = point) -> fragments; pre = h1 + D in the SWAPPED orientation (lane = feature 32 t + pl,
    // register r = point (r & 3) + 8 (r >> 2) + 4 h): per-feature constants become per-lane, sums over the points in-lane
    auto recompute = [&](u32x4 b0, f32x16 (&pre)[2]) {
        f32x16 h0a[2];
        u32x4 bf[NS][4];
        input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, h0a);          // gamma*h0n + beta
        split_fragment<true, NS, false, F16>(h0a, bf, a.negone);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float dsh = film[br * FILM_BR_FLOATS + 32 * t + pl];
#pragma unroll
            for (int r = 0; r < 16; ++r) pre[t][r] = dsh;
        }
        chain_mfma_swapped<NS, F16>(smem + L_PACK + PT_A1, br, lane, bf, pre);   // same products in the same order as the forward kernel
    };
    f32x16 pre[2];
    if constexpr (ROLES) {
        // the wave's one tile is recomputed HERE: it needs nothing from pass 1, and the role workgroups at the front of this
        // launch need ~1 us longer for the BN1-backward means than the staging above takes (same box: pass 2 16.3 -> 14.3 us at B = 8)
        __syncthreads();                                                   // weights, FiLM blocks, tables: in LDS
        recompute(input_fragment(h ? xb_t[0] : xa_t[0], h), pre);
        means_wait(a, mj, count, s12s, (int *)uns + 16);
    }
    __syncthreads();                                                       // weights, FiLM blocks, means, tables: all in LDS
    // BN1 backward per element:  dh1 = rstd1 (a dh2a - m1 - h1n m2) = C1 pa + ([pa > 0] ? G : C0),  C1 = -rstd1^2 m2,
    //   G[f][pt] = K1[f] do_a[pt] + K2[f] do_b[pt] + C0[f],  K_w = rstd1 a W2[w],  C0 = rstd1 (c/a m2 - m1)
    // G is an input-style contraction (two point values, a bias): its per-feature side as MFMA fragments, built once here
    if (threadIdx.x < 256) {
        const int b2 = threadIdx.x >> 7, t = (threadIdx.x >> 6) & 1, ln = threadIdx.x & 63, ff = 32 * t + (ln & 31), hh = ln >> 5;
        const float av = filmb[b2 * FB_BR + 0 * 64 + ff], rstd1 = filmb[b2 * FB_BR + 2 * 64 + ff], ca = filmb[b2 * FB_BR + 3 * 64 + ff];
        const float m1 = s12s[b2 * 128 + ff], m2 = s12s[b2 * 128 + 64 + ff];
        const float kw = rstd1 * av * w2s[b2 * 128 + 64 * hh + ff];
        *(u32x4 *)(smem + L_GFR + ((b2 * 2 + t) * 64 + ln) * 16) = input_weight_slots8_rne(kw, rstd1 * (ca * m2 - m1), hh);
    }
    lds_barrier();
    KP(2, 1)
#ifdef DPF_PROFILE
    unsigned long long tt[12];
    tt[11] = __builtin_amdgcn_s_memtime();
#endif
    // identity fragments: B operand of the MFMA that turns K = points fragments back into an accumulator with lane = point
    // (k-slot i of lane-half kg in k-step j holds point (i & 3) + 8 (2 j + (i >> 2)) + 4 kg, see kfrags_from_swapped)
    u32x4 eye[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int p0 = ((2 * d) & 3) + 8 * (2 * j2 + ((2 * d) >> 2)) + 4 * h, p1 = p0 + 1;
            eye[j2][d] = F16 ? ((pl == p0 ? 0x3C00u : 0u) | (pl == p1 ? 0x3C000000u : 0u))     // 1.0 as fp16 / as bf16
                             : ((pl == p0 ? 0x3F80u : 0u) | (pl == p1 ? 0x3F800000u : 0u));
        }
    float *pts = redw + wave * 2048;                                     // per-wave scratch (free until the reduction): [2][32] centred inputs
    const bool two_kept = a.kb >= 0;
    float *uplane = ubuf + (size_t)br * a.B * 2 * N;                       // this branch's plane of u_k
    // (the per-lane constants of the wave's branch are re-read from LDS where a tile needs them: held across the loop they
    // are eight registers of a kernel that sits at the 256 two waves per SIMD may have)
    const float winv = F16 ? u2f(__builtin_amdgcn_readfirstlane(f2u(*(const float *)(smem + L_PACK + pt_tail(NS) + 4 * br)))) : 1.0f;   // 2^-kw of the fp16 W1^T fragments
    f32x16 dw[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};       // dW1 of this branch over this wave's tiles [fo tile][fi tile]
    float rs[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};                  // [feature tile][k]: per-feature sums over this wave's tiles (both lane halves)
    int e_dw = 0;                                                          // F16: dw carries 2^e_dw * 16
#pragma unroll 1
    for (int ti = 0; ti < NT; ++ti) {
        TP(0)
        const int tile = PAIR ? (wave & 3) + 4 * ti : wave;
        const int tile0 = (blockIdx.x * TW + tile) * TILE;
        // (selects, not xa_t[ti]: a run-time index puts the array in scratch memory)
        const float xa = ti ? xa_t[NT - 1] : xa_t[0], xb = ti ? xb_t[NT - 1] : xb_t[0];
        const float doa = ti ? doa_t[NT - 1] : doa_t[0], dob = ti ? dob_t[NT - 1] : dob_t[0];
        const u32x4 b0 = input_fragment(h ? xb : xa, h);
        TP(1)
        // the centred inputs every lane needs in the swapped orientation (registers = points)
        if (!h) { pts[pl] = xa - ea; pts[32 + pl] = xb - eb; }
        if constexpr (!ROLES) recompute(b0, pre);
        TP(2)
        // ---- dh1 (in place), then as K = points fragments [feature tile][k-step]
        u32x4 xh[2][2], xl[2][2];
        float ginv = 1.0f;                                                 // F16: inverse of this tile's power-of-two scale
        {
            const u32x4 bd = input_fragment_rne(h ? dob : doa, h);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16 G = mfma(bd, *(const u32x4 *)(smem + L_GFR + ((br * 2 + t) * 64 + lane) * 16), zero16());
                const int fo = 32 * t + pl;                                 // this lane's feature (swapped orientation)
                const float rstd1 = filmb[br * FB_BR + 2 * 64 + fo], ca = filmb[br * FB_BR + 3 * 64 + fo];
                const float m1 = s12s[br * 128 + fo], m2 = s12s[br * 128 + 64 + fo];
                const float C1 = -(rstd1 * rstd1) * m2 * winv, C0 = rstd1 * (ca * m2 - m1);     // (pre is 2^kw pa for f16x3: C1 carries 2^-kw)
#pragma unroll
                for (int r = 0; r < 16; ++r) pre[t][r] = __builtin_fmaf(pre[t][r], C1, pre[t][r] > 0.f ? G[r] : C0);
            }
            if (tile0 + TILE > N) {              // ragged last tile of a cloud (wave-uniform, rare): no gradient from padding points
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pre[t][r] = tile0 + (r & 3) + 8 * (r >> 2) + 4 * h < N ? pre[t][r] : 0.f;
            }
            if constexpr (F16) {
                // fp16 operands need the values in fp16's range: the largest |dh1| of this wave's tile (64 features x 32 points)
                // goes to [2^13, 2^14) -- a power of two, so scaling and unscaling are exact and every element within 2^16 of the
                // largest keeps 22 significant bits
                float m = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(pre[t][r]));
                const uint32_t mb = wave_max_bits(m);                      // (a NaN anywhere may or may not surface here: its tile is lost either way)
                const int e = (int)((mb >> 23) & 0xFFu) - 127;
                const bool ok = mb != 0u && e > -100 && e < 100;           // (zero, denormal-tiny, Inf / NaN tiles: no scaling of their own)
                int e_t = ok ? 13 - e : (ti ? e_dw : 0);                   // this tile's scale 2^e_t
                if (ti) {
                    // dW1 accumulates over the wave's tiles in ONE set of accumulators: bring what is there to this tile's scale
                    // (exact, a power of two).  A later tile more than 2^60 below the earlier one keeps a smaller scale than
                    // its own (its products are then beneath the earlier tile's rounding anyway); scaling DOWN is not capped
                    // short of fp32's range, where the earlier tile's share is gone for the same reason.
                    e_t = e_t < e_dw + 60 ? e_t : e_dw + 60;
                    const int de = e_t - e_dw > -120 ? e_t - e_dw : -120;
                    if (de != 0) {
                        const float ratio = u2f((uint32_t)(127 + de) << 23);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) dw[mt][nt][r] *= ratio;
                    }
                }
                e_dw = e_t;
                ginv = u2f((uint32_t)(127 - e_t) << 23);
                kfrags_from_swapped_f16<false>(pre, u2f((uint32_t)(127 + e_t) << 23), a.negone, xh, xl);
            } else {
                kfrags_from_swapped<false>(pre, xh, xl);
            }
        }
        // The older wave of a SIMD wins the issue arbitration (r05 stamps: 9.8 K against 15.8 K ticks for the first tile): the
        // younger four get priority from the middle of their first tile on -- measured against the flip at the tile boundary
        // (+0.5 us) and earlier ones (+0.3 .. 0.5), tools/train_ab_prof.sh -- so that both halves of the workgroup reach the
        // reduction together
        if constexpr (PAIR) { if (ti == 0 && (wave >> 2)) __builtin_amdgcn_s_setprio(1); }
        TP(3)
        // ---- dh1 back to lane = point (an MFMA against the identity: hi + lo is exact) as the A fragments of dh0 = W1^T dh1
        u32x4 bg[2][4];
        {
            // r03: the hi and the lo parts are transposed SEPARATELY -- each result is a bf16 / fp16 value in an fp32 register, so
            // the fragments of the next contraction are one conversion per pair instead of a second symmetric split; 8 MFMAs
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                f32x16 dn[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4 (&xp)[2][2] = part ? xl : xh;
                    if constexpr (F16) {
                        dn[t] = mfma_f16(xp[t][0], eye[0], zero16());
                        dn[t] = mfma_f16(xp[t][1], eye[1], dn[t]);
                    } else {
                        dn[t] = mfma(xp[t][0], eye[0], zero16());
                        dn[t] = mfma(xp[t][1], eye[1], dn[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;          // as split_fragment
                        if constexpr (F16) bg[part][s][d] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(dn[t][r], dn[t][r + 1]));   // each value IS an fp16 number: exact
                        else bg[part][s][d] = pack_bf16_trunc(dn[t][r], dn[t][r + 1]);
                    }
            }
        }
        TP(4)
        // ---- dh0 = W1^T dh1 in the swapped orientation (lane = in-feature 32 t + pl, registers = the half's 16 points), relu
        // mask, then everything that hangs on it: the per-feature sums (in-lane adds), u_k (a sum over the FEATURES = over the
        // lanes: reduce_lanes16) and, below, dW1.  r01-r04 ran this contraction a second time with lane = point for u_k -- 24 + 2
        // MFMAs, a second mask, 128 FMAs fed by 16 dependent LDS reads of the coefficient table; the butterfly is 2 x 47 VALU.
        const float un = F16 ? ginv * winv : 1.0f;                         // dh0 carries the tile's scale and W1^T's (powers of two: exact)
        f32x16 h0s[2];                                                     // h0 pre-activation (x 16 for F16), lane = feature
        {
            input_mfma_swapped(smem + L_PACK + pt_a0(NS), br, lane, F16 ? input_fragment_x16(h ? xb : xa, h) : b0, h0s);
            f32x16 dh0s[2] = {zero16(), zero16()};
            chain_mfma_swapped<2, F16>(smem + L_PACK + pt_a1t(NS), br, lane, bg, dh0s);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float cka[2] = {cf[br * 128 + pl], cf[br * 128 + 32 + pl]}, ckb[2] = {cf[br * 128 + 64 + pl], cf[br * 128 + 96 + pl]};   // c_fk of this lane's features
            float ua, ub = 0.f;
            // (two code paths, chosen per wave: layers of pattern 1 keep ONE channel -- no x_b sum, no u_b)
            auto sums_and_u = [&](auto two) {
                constexpr bool TWO = decltype(two)::value;
                float wa[16], wb[16];                                      // c_fa dh0a, c_fb dh0a of this lane's two features, per point
                float sm[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
                for (int q = 0; q < 4; ++q) {                              // four of the half's points at a time: one 16-byte read per input
                    const f32x4 xa4 = *(const f32x4 *)(pts + 8 * q + h4);
                    f32x4 xb4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (TWO) xb4 = *(const f32x4 *)(pts + 32 + 8 * q + h4);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = 4 * q + i;
                            const float d = h0s[t][r] > 0.f ? dh0s[t][r] : 0.f;
                            sm[t][0] += d; sm[t][1] = __builtin_fmaf(d, xa4[i], sm[t][1]);
                            wa[r] = t ? __builtin_fmaf(cka[1], d, wa[r]) : cka[0] * d;
                            if constexpr (TWO) {
                                sm[t][2] = __builtin_fmaf(d, xb4[i], sm[t][2]);
                                wb[r] = t ? __builtin_fmaf(ckb[1], d, wb[r]) : ckb[0] * d;
                            }
                        }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < (TWO ? 3 : 2); ++k) rs[t][k] = __builtin_fmaf(sm[t][k], un, rs[t][k]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                           // the scratch is rewritten for the next tile
                // u_k[pt] = sum over this branch's features of c_fk dh0a[f][pt].  Lane i of each row ends up with point
                // (i & 3) + 8 ((i & 15) >> 2) + 4 h of the tile; plane `br`
                // (tried: the butterfly in the shadow of dW1's 24 MFMAs, which need nothing from it -- sched_group_barrier 1 : 4 --
                // the interleaved live ranges cost 130 registers of scratch at the 256 this kernel may use)
                ua = reduce_lanes16(wa, lane) * un;
                if constexpr (TWO) ub = reduce_lanes16(wb, lane) * un;
            };
            if (two_kept) sums_and_u(std::true_type{}); else sums_and_u(std::false_type{});
            const int pu = tile0 + (lane & 3) + 8 * ((lane & 15) >> 2) + 4 * h;
            if ((lane & 16) == 0 && pu < N) {
                uplane[((size_t)bi * 2 + 0) * N + pu] = ua;
                uplane[((size_t)bi * 2 + 1) * N + pu] = ub;
            }
        }
        TP(5)
        // ---- dW1[fo][fi] += sum_points dh1[fo][pt] * h0[fi][pt]: both operands are K = points fragments straight from
        // the swapped accumulators; hi/lo split like every other contraction: hi.hi, hi.lo, lo.hi
        {
            u32x4 yh[2][2], yl[2][2];
            // F16: relu(h0) x 16 as fp16 hi/lo (|BN0(h0)| < 2048 is what the f16x3 range monitor guarantees: 32768 < 65504)
            if constexpr (F16) kfrags_from_swapped_f16<true, false>(h0s, 1.0f, a.negone, yh, yl);
            else kfrags_from_swapped<true>(h0s, yh, yl);
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (F16) {
                            dw[mt][nt] = mfma_f16(xl[mt][j2], yh[nt][j2], dw[mt][nt]);
                            dw[mt][nt] = mfma_f16(xh[mt][j2], yl[nt][j2], dw[mt][nt]);
                            dw[mt][nt] = mfma_f16(xh[mt][j2], yh[nt][j2], dw[mt][nt]);
                        } else {
                            dw[mt][nt] = mfma(xl[mt][j2], yh[nt][j2], dw[mt][nt]);
                            dw[mt][nt] = mfma(xh[mt][j2], yl[nt][j2], dw[mt][nt]);
                            dw[mt][nt] = mfma(xh[mt][j2], yh[nt][j2], dw[mt][nt]);
                        }
                    }
        }
        TP(6)
    }
    // ---- the one workgroup reduction: dW1 through per-wave LDS slots (plain stores: LDS float atomics are ~1000 cycles per wave
    // instruction), two rounds of 32 accumulator registers, then the three per-feature sums.  16-byte LDS accesses: a lane stores
    // its registers 4g .. 4g+3 as one quad, a thread sums ONE quad over the branch's waves in wave order, each wave's share
    // times the power of two that takes it back to the true scale (an FMA where r04 had a multiply per element and an add)
    KP(2, 2)
    if (lane == 0) uns[wave] = F16 ? u2f((uint32_t)(127 - e_dw - 4) << 23) : 1.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 3; ++k) rs[t][k] = half_sum(rs[t][k]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        // (lds_barrier, not __syncthreads: that one waits for every outstanding vector-memory operation, i.e. for the PREVIOUS round's
        // 16 global stores per thread -- r05 stamps: 6 K ticks from the last tile to the exit, most of them two store round trips)
        lds_barrier();                                                       // every wave is through with its scratch / the previous round has been read
#ifdef DPF_PROFILE
        if (mt == 0) { __builtin_amdgcn_sched_barrier(0); tt[9] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
        f32x4 *slot = (f32x4 *)(redw + wave * 2048) + lane;                // quad (g, lane) at [g * 64 + lane]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {dw[mt][nt][4 * g + 0], dw[mt][nt][4 * g + 1], dw[mt][nt][4 * g + 2], dw[mt][nt][4 * g + 3]};
                slot[(4 * nt + g) * 64] = v;
            }
        lds_barrier();    
        const int e = threadIdx.x;                                         // quad index: (nt, g, lane)
        const int ln = e & 63, g = (e >> 6) & 3, nt = e >> 8;
        const int fo = 32 * mt + 8 * g + 4 * (ln >> 5), fi = 32 * nt + (ln & 31);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int w0 = PAIR ? 4 * b : 0;
            f32x4 t = *((const f32x4 *)(redw + w0 * 2048) + e);
            if constexpr (F16) { const float u = uns[w0]; t.x *= u; t.y *= u; t.z *= u; t.w *= u; }
#pragma unroll
            for (int w = 1; w < NWB; ++w) {
                const f32x4 v = *((const f32x4 *)(redw + (w0 + w) * 2048) + e);
                if constexpr (F16) {
                    const float u = uns[w0 + w];
                    t.x = __builtin_fmaf(v.x, u, t.x); t.y = __builtin_fmaf(v.y, u, t.y); t.z = __builtin_fmaf(v.z, u, t.z); t.w = __builtin_fmaf(v.w, u, t.w);
                } else {
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
            }
            float *o = part2 + (blk * 2 + (PAIR ? b : (int)blockIdx.z)) * P2_J + P2_W;
            o[(fo + 0) * 64 + fi] = t.x;
            o[(fo + 1) * 64 + fi] = t.y;
            o[(fo + 2) * 64 + fi] = t.z;
            o[(fo + 3) * 64 + fi] = t.w;
        }
    }
    lds_barrier();
#ifdef DPF_PROFILE
    { __builtin_amdgcn_sched_barrier(0); tt[10] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
    if (!h) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 3; ++k) redw[wave * 256 + k * 64 + 32 * t + pl] = rs[t][k];
    }
    lds_barrier();    
    if (threadIdx.x < NB * 192) {
        const int b = threadIdx.x / 192, i = threadIdx.x - 192 * b, w0 = PAIR ? 4 * b : 0;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWB; ++w) t += redw[(w0 + w) * 256 + i];
        part2[(blk * 2 + (PAIR ? b : (int)blockIdx.z)) * P2_J + (i < 64 ? P2_S + i : P2_A - 64 + i)] = t;
    }
    KP(2, 3)
#ifdef DPF_PROFILE
    if (g_tprof != nullptr && lane == 0 && blk < 2) {
        for (int i = 0; i < 12; ++i) g_tprof[(blk * TW + wave) * 14 + i] = tt[i];
        g_tprof[(blk * TW + wave) * 14 + 12] = t_entry;
        g_tprof[(blk * TW + wave) * 14 + 13] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// (tbwd3f_kernel below) Totals of pass 2 -> d gamma0, d beta0, dW1, dW0 and the coefficients of the input gradient.
//   tot[br][P2_J] doubles.  BN0 backward:  dh0pre = rstd0*gamma0*(dh0a - A - h0n*Bc),  A = S/P,  Bc = Sg/P
//   dW0[f][k] = sum_pt dh0pre * x_k = rstd0*gamma0*(Sk - A*sum x_k - Bc * sum h0n*x_k), and h0n is linear in x:
//   sum_pt h0n_f x_k / P = rstd0_f (w_fa cov(a,k) + w_fb cov(b,k)) + (mean of h0n = 0) * E[x_k]
//   dx_k[pt] = u_k[pt] - C_k - alpha_k (x_a - E x_a) - beta_k (x_b - E x_b):  coef[k] = {C_k, alpha_k, beta_k, E x_k}
//   (r05: centred -- the r01-r04 form alpha_k x_a + beta_k x_b - delta_k cancelled delta_k = sum_f c_fk Bc_f rstd0_f mean0_f against
//   the products per point, in fp32; h0n = rstd0 (w_a (x_a - E x_a) + w_b (x_b - E x_b)) has no such term)

// Pass 3: the conditioner path of d(input points), elementwise
// tfinish2 + tbwd3 in one launch (a tiny dependent kernel costs ~4.5 us here whatever it does): EVERY workgroup
// recomputes the six input-gradient coefficients from the 512 totals (128 threads of double arithmetic, same
// operations in the same order as tfinish2, so the same bits), workgroup (0,0) also writes d gamma0 / d beta0 / dW0,
// and the dW1 totals are converted by whoever comes first (grid-stride).  No cross-workgroup dependency.
__global__ __launch_bounds__(256) void tbwd3f_kernel(int N, int ka, int kb, int nk, double count, const double *__restrict__ tot,
                                                     const float *__restrict__ tcanon_l, const float *__restrict__ stats_l,
                                                     float *__restrict__ dcanon_l, const float *__restrict__ p_in,
                                                     const float *__restrict__ ubuf, const float *__restrict__ ubuf2,
                                                     float *__restrict__ dp_in) {
    __shared__ double acc[2][8];
    __shared__ float coef[8];
    bwd3_coefs(bwd3_loads(nk, tot, tcanon_l, stats_l), blockIdx.y * gridDim.x + blockIdx.x, nk, count, dcanon_l, acc, coef);
    const int bi = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float *pc = p_in + (size_t)bi * 3 * N;
    float *d = dp_in + (size_t)bi * 3 * N;
    const float xa = pc[(size_t)ka * N + n], xb = kb >= 0 ? pc[(size_t)kb * N + n] : 0.f;
    float ua = ubuf[((size_t)bi * 2 + 0) * N + n], ub = kb >= 0 ? ubuf[((size_t)bi * 2 + 1) * N + n] : 0.f;
    if (ubuf2 != nullptr) {            // branch-split pass 2: the mu branch's plane (as in tbwd1_kernel's prologue)
        ua = __fadd_rn(ua, ubuf2[((size_t)bi * 2 + 0) * N + n]);
        if (kb >= 0) ub = __fadd_rn(ub, ubuf2[((size_t)bi * 2 + 1) * N + n]);
    }
    const float xac = __fsub_rn(xa, coef[3]), xbc = __fsub_rn(xb, coef[7]);
    d[(size_t)ka * N + n] = __fadd_rn(d[(size_t)ka * N + n], cond_path(ua, coef[0], coef[1], coef[2], xac, xbc));
    if (kb >= 0)
        d[(size_t)kb * N + n] = __fadd_rn(d[(size_t)kb * N + n], cond_path(ub, coef[4], coef[5], coef[6], xac, xbc));
}


#ifdef DPF_PROFILE
__global__ void tprof_set_kernel(unsigned long long *p) { g_tprof = p; }
#endif

}  // namespace

// operand parts of the forward contraction; DPF_PREC_F16X3 shares the hi/lo layout of bf16x3 with fp16 values (r03)
static inline int t_ns(int precision) {
    return (precision == DPF_PREC_BF16X3 || precision == DPF_PREC_F16X3) ? 2 : (precision == DPF_PREC_BF16X6 ? 3 : 0);
}

extern "C" size_t dpf_flow_train_canon_floats(void) { return (size_t)T_LAYER; }
extern "C" size_t dpf_flow_train_packed_bytes(int n_layers, int precision) {
    return t_ns(precision) ? (size_t)n_layers * pt_bytes(t_ns(precision)) : 0;
}
extern "C" size_t dpf_flow_train_stats_floats(void) { return (size_t)ST_LAYER; }
extern "C" size_t dpf_flow_train_film_floats(int B) { return (size_t)B * (512 + FB_CLOUD); }

static inline int t_nblk(int B, int N) { return B * ((N + TBLK - 1) / TBLK); }

struct TWork {
    double *xpart, *sums, *tot2;
    float *part1, *pc, *s12, *part2, *dout, *ubuf, *coef;
    unsigned *tickets;           // the arrival counters of the role workgroups: word B = column sums in pass 1, word B + 1 = means in pass 2 (words [0, B): unused since r05)
};
static size_t carve(void *ws, int B, int N, TWork *w) {
    const size_t nblk = (size_t)t_nblk(B, N), nbx = (size_t)B * ((N + TILE - 1) / TILE);   // the flow kernel's smallest workgroup is one tile
    uint8_t *p = (uint8_t *)ws;
    auto take = [&](size_t bytes) { uint8_t *q = p; p += (bytes + 255) / 256 * 256; return q; };
    uint8_t *xpart = take(nbx * 8 * sizeof(double));
    uint8_t *sums = take(256 * sizeof(double));
    uint8_t *tot2 = take(2 * P2_J * sizeof(double));
    uint8_t *part1 = take(nblk * 520 * 4);
    uint8_t *pc = take((size_t)B * 520 * 4);
    uint8_t *tickets = take((size_t)(B + 16) * 4);                      // + the arrival counter of the fused column sums (word B)
    uint8_t *s12 = take(256 * 4);
    uint8_t *part2 = take(nblk * 2 * P2_J * 4);
    uint8_t *dout = take((size_t)B * 4 * N * 4);
    uint8_t *ubuf = take((size_t)2 * B * 2 * N * 4);                     // two planes (branch-split pass 2 writes one per branch)
    uint8_t *coef = take(8 * 4);
    if (w) {
        w->xpart = (double *)xpart; w->sums = (double *)sums; w->tot2 = (double *)tot2; w->part1 = (float *)part1;
        w->tickets = (unsigned *)tickets;
        w->pc = (float *)pc; w->s12 = (float *)s12; w->part2 = (float *)part2; w->dout = (float *)dout;
        w->ubuf = (float *)ubuf; w->coef = (float *)coef;
    }
    return (size_t)(p - (uint8_t *)ws);
}
// workspace for one layer call, forward or backward (contents do not survive the call)
extern "C" size_t dpf_flow_train_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return carve(nullptr, B, N, nullptr) + 256;
}

extern "C" int dpf_flow_train_pack(int n_layers, int precision, const float *tcanon, void *packed, dpf_stream_t stream) {
    const int ns = t_ns(precision);
    if (n_layers <= 0 || !tcanon || !packed) return DPF_EINVAL;
    if (!ns) return DPF_ENOSUP;
    if (precision == DPF_PREC_F16X3) hipLaunchKernelGGL((tpack_kernel<2, true>), dim3(n_layers), dim3(256), 0, (hipStream_t)stream, tcanon, (uint8_t *)packed);
    else if (ns == 2) hipLaunchKernelGGL(tpack_kernel<2>, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, tcanon, (uint8_t *)packed);
    else hipLaunchKernelGGL(tpack_kernel<3>, dim3(n_layers), dim3(256), 0, (hipStream_t)stream, tcanon, (uint8_t *)packed);
    return (int)hipGetLastError();
}

// ---- per-kernel timing of the training stack (diagnostics: bench.py's `roofline.kernels`) ---------------------------
// dpf_train_kernel_times(1, ...) switches it on: every launch of the per-layer kernels is bracketed by two HIP events on the
// launch stream (graph recording / replay is bypassed while it is on); (0, us, calls) synchronises, returns the summed
// event-to-event time (us) and the number of launches per kernel id, and switches it off.  ids: 0 tstats_x, 1 tstats_h1,
// 2 tfold, 3 the layer's flow_kernel (L = 1), 4 tbwd1, 5 tbwd2, 6 tcolsum, 7 tbwd3f.
namespace {
constexpr int KT_N = 8;
struct KTimer {
    std::atomic<int> on{0};
    std::mutex mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[KT_N];
    hipEvent_t begin(int id, hipStream_t s) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        (void)hipEventRecord(a, s);
        std::lock_guard<std::mutex> lock(mu);
        ev[id].push_back({a, b});
        return b;
    }
};
KTimer &ktimer() { static KTimer t; return t; }
struct KScope {       // brackets the launches issued during its lifetime
    hipEvent_t stop = nullptr;
    hipStream_t s;
    KScope(int id, hipStream_t st) : s(st) { if (ktimer().on.load(std::memory_order_relaxed)) stop = ktimer().begin(id, st); }
    ~KScope() { if (stop) (void)hipEventRecord(stop, s); }
};
}  // namespace

extern "C" int dpf_train_kernel_times(int enable, double *us_out, long *calls_out) {
    KTimer &t = ktimer();
    static int graph_was = 1;
    if (enable) {
        std::lock_guard<std::mutex> lock(t.mu);
        for (auto &v : t.ev) { for (auto &p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); } v.clear(); }
        if (!t.on.load()) graph_was = dpf_graph_enabled_flag().exchange(0);      // events cannot sit inside a replayed graph
        t.on.store(1);
        return 0;
    }
    if (t.on.exchange(0)) dpf_graph_enabled_flag().store(graph_was);
    if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(t.mu);
    for (int i = 0; i < KT_N; ++i) {
        double us = 0;
        for (auto &p : t.ev[i]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) us += 1e3 * ms;
            (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
        }
        if (us_out) us_out[i] = us;
        if (calls_out) calls_out[i] = (long)t.ev[i].size();
        t.ev[i].clear();
    }
    return 0;
}

template <int NS, bool F16 = false>
static int prepare_layer(int B, int N, int ka, int kb, const float *tcanon_l, void *packed_l, const float *fm_l,
                         const float *p_in, float *stats_l, float *film_l, float flow_eps, void *workspace, hipStream_t s,
                         int xrows) {
    TWork w;
    carve(workspace, B, N, &w);
    // moments of the layer's input: left behind by the previous layer's flow kernel (xrows partial rows per cloud), or
    // computed here for the first layer of the call
    const int nbx = xrows > 0 ? xrows : (N + 255) / 256;
    const double count = (double)B * N;
    if (xrows <= 0) { KScope ks(0, s); hipLaunchKernelGGL(tstats_x_kernel, dim3(nbx, B), dim3(256), 0, s, N, ka, kb, p_in, w.xpart); }
    TArgs a;
    a.packed_l = (const uint8_t *)packed_l; a.tcanon_l = tcanon_l; a.film_l = film_l; a.filmb_l = film_l + (size_t)B * 512;
    a.stats_l = stats_l; a.p_in = p_in; a.B = B; a.N = N; a.ka = ka; a.kb = kb; a.wa = 0; a.wb = 0; a.mode = 0;
    a.eps = flow_eps; a.negone = -1.0f;
    static LdsLimit lim_h1;
    if (hipError_t e = lim_h1.ensure((const void *)tstats_h1_kernel<NS, F16>, pt_a0n(NS)); e != hipSuccess) return (int)e;
    const dim3 grid((N + TBLK - 1) / TBLK, B);
    static const int split_env = getenv("DPF_TRAIN_SPLIT") ? atoi(getenv("DPF_TRAIN_SPLIT")) : -1;
    // one branch per workgroup for small batches (r04, B = 8: 9.5 -> 8.7 us; at 128 workgroups -- B = 16 -- passes 1 and the
    // statistics are better off unsplit, only pass 2 gains)
    const bool split = split_env >= 0 ? split_env != 0 : (int)(grid.x * grid.y) <= 64;
    { KScope ks(1, s);
    hipLaunchKernelGGL((tstats_h1_kernel<NS, F16>), dim3(grid.x, grid.y, split ? 2 : 1), dim3(TW * 64), pt_a0n(NS), s, a, w.part1, nbx * B, count, w.xpart,
                       (uint8_t *)packed_l + pt_a0(NS)); }
    { KScope ks(2, s);
    hipLaunchKernelGGL(tfold_kernel, dim3(8), dim3(1024), 0, s, count, (int)(grid.x * grid.y), w.part1, tcanon_l, fm_l, B, flow_eps,
                       stats_l, film_l, film_l + (size_t)B * 512, F16 ? (const float *)((const uint8_t *)packed_l + pt_tail(NS)) : nullptr); }
    return (int)hipGetLastError();
}

// Training-mode forward of an L-layer stack: per layer (in the order the mode prescribes) the batch statistics
// and folds, then the layer itself through dpf_flow_forward(n_layers = 1).
static int flow_train_forward_direct(int n_layers, int B, int N, int mode, int precision, const int *meta_host,
                                     const int *meta_dev, const float *tcanon, void *packed, const float *fm,
                                     const float *p_in, float *ps, float *mus, float *logvars, float *stats, float *film,
                                     float flow_eps, void *workspace, dpf_stream_t stream) {
    if (n_layers <= 0 || B <= 0 || N <= 0 || !meta_host || !meta_dev || !tcanon || !packed || !fm || !p_in || !ps || !mus ||
        !logvars || !stats || !film || !workspace)
        return DPF_EINVAL;
    if (mode != DPF_MODE_DIRECT && mode != DPF_MODE_INVERSE) return DPF_EINVAL;
    const int ns = t_ns(precision);
    if (!ns || B > 65535) return DPF_ENOSUP;
    const size_t lst = (size_t)B * 3 * N, fls = dpf_flow_train_film_floats(B), fms = (size_t)4 * B * DPF_FLOW_F;
    const float *cur = p_in;
    TWork w;
    carve(workspace, B, N, &w);
    int xrows = 0;                                   // partial rows per cloud the previous layer's kernel left in w.xpart
    for (int step = 0; step < n_layers; ++step) {
        const int l = mode == DPF_MODE_DIRECT ? step : n_layers - 1 - step;
        const int *m = meta_host + 4 * l;
        uint8_t *pk = (uint8_t *)packed + (size_t)l * pt_bytes(ns);
        float *film_l = film + l * fls;
#define DPF_PREP(...)                                                                                                    \
    prepare_layer<__VA_ARGS__>(B, N, m[0], m[1], tcanon + (size_t)l * T_LAYER, pk, fm + l * fms, cur,                    \
                               stats + (size_t)l * ST_LAYER, film_l, flow_eps, workspace, (hipStream_t)stream, xrows)
        int rc = precision == DPF_PREC_F16X3 ? DPF_PREP(2, true) : (ns == 2 ? DPF_PREP(2) : DPF_PREP(3));
#undef DPF_PREP
        if (rc) return rc;
        // the layer itself; its epilogue leaves the moments of the NEXT layer's kept coordinates (w.xpart was consumed by
        // this layer's tstats_h1 above)
        xrows = 0;
        KScope ks_flow(3, (hipStream_t)stream);
        if (step + 1 < n_layers) {
            const int *mnext = meta_host + 4 * (mode == DPF_MODE_DIRECT ? l + 1 : l - 1);
            rc = flow_forward_xstats(B, N, mode, precision, pk, meta_dev + 4 * l, film_l, cur, ps + l * lst, mus + l * lst,
                                     logvars + l * lst, flow_eps, stream, w.xpart, mnext[0], mnext[1], &xrows);
        } else {
            // (the csrc-internal entry, not dpf_flow_forward: `pk` is tpack_kernel's 32-point-tile block -- the public entry
            // would hand a small batch to the 16-point-tile kernel, whose fragments dpf_flow_pack lays out)
            rc = flow_forward_xstats(B, N, mode, precision, pk, meta_dev + 4 * l, film_l, cur, ps + l * lst, mus + l * lst,
                                     logvars + l * lst, flow_eps, stream, nullptr, 0, -1, nullptr);
        }
        if (rc) return rc;
        cur = ps + l * lst;
    }
    return 0;
}

// The stack's ~6 n_layers launches as one graph launch once the same call has been seen twice (graph_cache.h)
extern "C" int dpf_flow_train_forward(int n_layers, int B, int N, int mode, int precision, const int *meta_host,
                                      const int *meta_dev, const float *tcanon, void *packed, const float *fm,
                                      const float *p_in, float *ps, float *mus, float *logvars, float *stats, float *film,
                                      float flow_eps, void *workspace, dpf_stream_t stream) {
    auto direct = [&](hipStream_t st) {
        return flow_train_forward_direct(n_layers, B, N, mode, precision, meta_host, meta_dev, tcanon, packed, fm, p_in, ps, mus,
                                         logvars, stats, film, flow_eps, workspace, (dpf_stream_t)st);
    };
    if (n_layers <= 0 || !meta_host) return direct((hipStream_t)stream);
    static GraphCache cache;
    GraphKey k;
    k.val(n_layers); k.val(B); k.val(N); k.val(mode); k.val(precision); k.add(meta_host, sizeof(int) * 4 * n_layers);
    k.val(meta_dev); k.val(tcanon); k.val(packed); k.val(fm); k.val(p_in); k.val(ps); k.val(mus); k.val(logvars); k.val(stats);
    k.val(film); k.val(flow_eps); k.val(workspace);
    int dev = 0;
    (void)hipGetDevice(&dev);
    k.val(dev);
    return cache.run(k, (hipStream_t)stream, direct);
}

template <int NS, bool F16 = false>
static int backward_layer(int B, int N, int mode, int ka, int kb, int wa, int wb, const float *tcanon_l, const void *packed_l,
                          const float *film_l, const float *stats_l, const float *p_in, const float *mu_l, const float *lv_l,
                          const float *g_p, const float *g_p2, const float *g_mu, const float *g_lv, float *dp_in, float *dcanon_l,
                          float *dfm_l, float flow_eps, void *workspace, hipStream_t s, PrevLayer *pv, int *pass2_launches, bool last) {
    TWork w;
    carve(workspace, B, N, &w);
    TArgs a;
    a.packed_l = (const uint8_t *)packed_l; a.tcanon_l = tcanon_l; a.film_l = film_l; a.filmb_l = film_l + (size_t)B * 512;
    a.stats_l = stats_l; a.p_in = p_in; a.B = B; a.N = N; a.ka = ka; a.kb = kb; a.wa = wa; a.wb = wb; a.mode = mode;
    a.eps = flow_eps; a.negone = -1.0f;
    const dim3 grid((N + TBLK - 1) / TBLK, B);
    const int nblk = grid.x * grid.y;
    const double count = (double)B * N;
    const int lds1 = pt_a0n(NS) + 4096 + (TW * 520 + 256 + TW * 64 + 8 + 24) * 4, lds2 = l_red(NS) + 4096 + TW * XY_WAVE * 2;
    static LdsLimit lim_b1;
    if (hipError_t e = lim_b1.ensure((const void *)tbwd1_kernel<NS, F16>, lds1); e != hipSuccess) return (int)e;
    static LdsLimit lim_b1s;
    if (hipError_t e = lim_b1s.ensure((const void *)tbwd1_kernel<NS, F16, true>, lds1); e != hipSuccess) return (int)e;
    // r04: the column sums of the layer above's pass-2 partials ride in this launch (ColsumJob) instead of a tcolsum launch
    // of their own between the two layers; DPF_TRAIN_FUSE_COLSUM=0 keeps the separate launch
    static const int fuse_env = getenv("DPF_TRAIN_FUSE_COLSUM") ? atoi(getenv("DPF_TRAIN_FUSE_COLSUM")) : 1;
    ColsumJob cs = {};
    dim3 grid1 = grid;
    if (fuse_env && pv->has) {
        cs.part2 = w.part2; cs.tot = w.tot2; cs.dcanon_prev = pv->dcanon_l; cs.flag = w.tickets + B; cs.nrows = nblk;
        cs.target = (unsigned)CS_CRIT * (unsigned)(++pv->fused_launches);
        cs.rows = (CS_CRIT + (int)grid.x - 1) / (int)grid.x;
        grid1.y = cs.rows + B + (2 * P2_J / 32 - CS_CRIT / 2 + grid.x - 1) / grid.x;
    }
    // small batches: the two branches of passes 1 and 2 in two workgroups each (at most half a workgroup per CU otherwise)
    static const int split_env = getenv("DPF_TRAIN_SPLIT") ? atoi(getenv("DPF_TRAIN_SPLIT")) : -1;
    // (bf16x6's three forward parts do not leave the one-branch-per-wave form its registers: that precision keeps one branch per workgroup)
    const bool split2 = NS == 3 || (split_env >= 0 ? split_env != 0 : nblk <= 128);
    const bool split1 = split_env >= 0 ? split_env != 0 : nblk <= 64;      // (B = 16: tbwd1 15.7 us unsplit, 18.1 split)
    grid1.z = split1 ? 2 : 1;
    // Who finishes pass 1 -- per-cloud totals, FiLM gradients, dW2 / db2, the BN1-backward means?  Pass 2 needs a CU per workgroup
    // (158 KB of LDS): role workgroups at the front of its grid (MeansJob) cost nothing where CUs are idle and a whole round of
    // late workgroups where they are not.  So: roles while the ordinary workgroups + MJ_ROLES fit the chip's CUs, else pass 1's
    // per-cloud ticket and a recomputation of the means by every workgroup of pass 2 (r02-r04).  DPF_TRAIN_ROLES=0/1 forces either.
    static const int roles_env = getenv("DPF_TRAIN_ROLES") ? atoi(getenv("DPF_TRAIN_ROLES")) : -1;
    static const int n_cu = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int wg2 = nblk * (split2 ? 2 : 1);
    // (the role form is built for the one-branch-per-workgroup kernel only: that is the form small batches run)
    const bool roles = split2 && (roles_env >= 0 ? roles_env != 0 : wg2 + MJ_ROLES <= n_cu);
    { KScope ks(4, s);
    unsigned *tk = roles ? nullptr : w.tickets;
    if (split1)
        hipLaunchKernelGGL((tbwd1_kernel<NS, F16, true>), grid1, dim3(TW * 64), lds1, s, a, g_p, g_p2, g_mu, g_lv, mu_l, lv_l, dp_in, w.dout, w.part1, *pv, tk, w.pc, dfm_l, cs);
    else
        hipLaunchKernelGGL((tbwd1_kernel<NS, F16>), grid1, dim3(TW * 64), lds1, s, a, g_p, g_p2, g_mu, g_lv, mu_l, lv_l, dp_in, w.dout, w.part1, *pv, tk, w.pc, dfm_l, cs); }
    MeansJob mj;
    mj.part1 = w.part1; mj.pc = w.pc; mj.dfm_l = dfm_l; mj.s12 = w.s12; mj.flag = w.tickets + B + 1; mj.nb = (int)grid.x;
    mj.target = 0; mj.rows = 0;
    if (roles) {
        mj.target = (unsigned)MJ_ROLES * (unsigned)(++*pass2_launches);
        mj.rows = (MJ_ROLES + (int)grid.x - 1) / (int)grid.x;
    }
    const dim3 grid2(grid.x, mj.rows + grid.y, split2 ? 2 : 1);
    // split2: one branch per workgroup, a tile per wave; else (two-part precisions): one branch per wave, two tiles per wave
#define DPF_P2(PAIRV, ROLESV, LDSV)                                                                                                  \
    {                                                                                                                               \
        static LdsLimit lim;                                                                                                        \
        if (hipError_t e = lim.ensure((const void *)tbwd2_kernel<NS, F16, PAIRV, ROLESV>, LDSV); e != hipSuccess) return (int)e;    \
        KScope ks(5, s);                                                                                                            \
        hipLaunchKernelGGL((tbwd2_kernel<NS, F16, PAIRV, ROLESV>), grid2, dim3(TW * 64), LDSV, s, a, mj, count, dcanon_l, w.dout, w.ubuf, w.part2); \
    }
    if (split2) { if (roles) DPF_P2(false, true, lds2) else DPF_P2(false, false, lds2) }
    else if constexpr (NS == 2) DPF_P2(true, false, lds2)
#undef DPF_P2
    const float *ubuf2 = w.ubuf + (size_t)B * 2 * N;                      // u_k comes in two planes (one per branch) either way
    if (!fuse_env || last) {              // (fused: the next backward layer's pass 1 sums these partials; the last layer has none)
        KScope ks(6, s);
        hipLaunchKernelGGL(tcolsum_kernel, dim3((2 * P2_J + 31) / 32), dim3(1024), 0, s, nblk, 2 * P2_J, w.part2, w.tot2, dcanon_l, P2_J);
    }
    // pass 3 (the conditioner path of d(input points), d gamma0 / d beta0 / dW0 / dW1 from the totals): folded into the NEXT
    // backward layer's pass 1, which needs that gradient anyway; only the last layer of the call launches it
    if (last) {
        KScope ks(7, s);
        hipLaunchKernelGGL(tbwd3f_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, N, ka, kb, kb >= 0 ? 2 : 1, count, w.tot2,
                           tcanon_l, stats_l, dcanon_l, p_in, w.ubuf, ubuf2, dp_in);
    }
    pv->tot = w.tot2; pv->tcanon_l = tcanon_l; pv->stats_l = stats_l; pv->ubuf = w.ubuf; pv->ubuf2 = ubuf2; pv->x = p_in; pv->dcanon_l = dcanon_l;
    pv->count = count; pv->ka = ka; pv->kb = kb; pv->has = 1;
    return (int)hipGetLastError();
}

// Backward of the stack, layers in the reverse of the forward order.  The gradient that reaches layer l's p_out is
// g_p(l) plus what the next layer passes down; g_p / g_mu / g_lv of a layer may be NULL (zero).  dp_in (B,3,N),
// dcanon (L, T_LAYER) and dfm (L,[br][sub][B][64]) are fully overwritten; dp_tmp is a (B,3,N) scratch buffer.
template <class GP>
static int backward_stack(int n_layers, int B, int N, int mode, int precision, const int *meta_host, const float *tcanon,
                          const void *packed, const float *film, const float *stats, const float *p_in, const float *ps,
                          const float *mus, const float *logvars, GP &&grad_of, float *dp_in, float *dp_tmp, float *dcanon,
                          float *dfm, float flow_eps, void *workspace, dpf_stream_t stream) {
    if (n_layers <= 0 || B <= 0 || N <= 0 || !meta_host || !tcanon || !packed || !film || !stats || !p_in || !ps || !mus || !logvars ||
        !dp_in || !dp_tmp || !dcanon || !dfm || !workspace)
        return DPF_EINVAL;
    if (mode != DPF_MODE_DIRECT && mode != DPF_MODE_INVERSE) return DPF_EINVAL;
    const int ns = t_ns(precision);
    if (!ns || B > 65535) return DPF_ENOSUP;
    const size_t lst = (size_t)B * 3 * N, fls = dpf_flow_train_film_floats(B), fms = (size_t)4 * B * DPF_FLOW_F;
    const float *chain = nullptr;
    int pass2_launches = 0;          // pass-2 launches of this call that carried role workgroups so far (their counter is monotonic over the call)
    PrevLayer pv = {};
    {   // no layer above the first one (has = 0), but pointers its pass 1 can load from unconditionally
        TWork w0;
        carve(workspace, B, N, &w0);
        const int l0 = mode == DPF_MODE_DIRECT ? n_layers - 1 : 0;
        pv.tot = w0.tot2; pv.tcanon_l = tcanon + (size_t)l0 * T_LAYER; pv.stats_l = stats + (size_t)l0 * ST_LAYER;
        pv.ka = 0; pv.kb = -1;
    }
    {   // the role workgroups' arrival counters start at zero (they are monotonic over the call).  A fill KERNEL: as a captured
        // memset node the clear was not reliably ordered before the first pass-1 kernel of a replay (zero_fill.h)
        TWork w;
        carve(workspace, B, N, &w);
        if (hipError_t e = dpf_zero_async(w.tickets, (size_t)(B + 16) * 4, (hipStream_t)stream); e != hipSuccess) return (int)e;
    }
    for (int step = n_layers - 1; step >= 0; --step) {
        const int l = mode == DPF_MODE_DIRECT ? step : n_layers - 1 - step;
        const int lprev = mode == DPF_MODE_DIRECT ? step - 1 : n_layers - step;       // layer whose output fed layer l
        const float *pin = step == 0 ? p_in : ps + lprev * lst;
        float *out = (step & 1) ? dp_tmp : dp_in;                                      // step 0 writes dp_in
        const int *m = meta_host + 4 * l;
#define DPF_BWD(NSV, ...)                                                                                                 \
    backward_layer<NSV, ##__VA_ARGS__>(B, N, mode, m[0], m[1], m[2], m[3], tcanon + (size_t)l * T_LAYER,                \
                        (const uint8_t *)packed + (size_t)l * pt_bytes(NSV), film + l * fls, stats + (size_t)l * ST_LAYER, pin, \
                        mus + l * lst, logvars + l * lst, grad_of(0, l), chain, grad_of(1, l), grad_of(2, l), out,       \
                        dcanon + (size_t)l * T_LAYER, dfm + l * fms, flow_eps, workspace, (hipStream_t)stream, &pv, &pass2_launches, step == 0)
        const int rc = precision == DPF_PREC_F16X3 ? DPF_BWD(2, true) : (ns == 2 ? DPF_BWD(2) : DPF_BWD(3));
#undef DPF_BWD
        if (rc) return rc;
        chain = out;
    }
    return 0;
}

// g_ps / g_mus / g_lvs: (L,B,3,N) gradients w.r.t. the three output lists (g_mus / g_lvs may be NULL)
extern "C" int dpf_flow_train_backward(int n_layers, int B, int N, int mode, int precision, const int *meta_host,
                                       const float *tcanon, const void *packed, const float *film, const float *stats,
                                       const float *p_in, const float *ps, const float *mus, const float *logvars,
                                       const float *g_ps, const float *g_mus, const float *g_lvs, float *dp_in, float *dp_tmp,
                                       float *dcanon, float *dfm, float flow_eps, void *workspace, dpf_stream_t stream) {
    if (!g_ps) return DPF_EINVAL;
    const size_t lst = (size_t)(B > 0 ? B : 0) * 3 * (N > 0 ? N : 0);
    const float *base[3] = {g_ps, g_mus, g_lvs};
    auto direct = [&](hipStream_t st) {
        return backward_stack(n_layers, B, N, mode, precision, meta_host, tcanon, packed, film, stats, p_in, ps, mus, logvars,
                              [&](int which, int l) { return base[which] ? base[which] + l * lst : nullptr; }, dp_in, dp_tmp,
                              dcanon, dfm, flow_eps, workspace, (dpf_stream_t)st);
    };
    if (n_layers <= 0 || !meta_host) return direct((hipStream_t)stream);
    static GraphCache cache;
    GraphKey k;
    k.val(n_layers); k.val(B); k.val(N); k.val(mode); k.val(precision); k.add(meta_host, sizeof(int) * 4 * n_layers);
    k.val(tcanon); k.val(packed); k.val(film); k.val(stats); k.val(p_in); k.val(ps); k.val(mus); k.val(logvars); k.val(g_ps);
    k.val(g_mus); k.val(g_lvs);
    k.val(dp_in); k.val(dp_tmp); k.val(dcanon); k.val(dfm); k.val(flow_eps); k.val(workspace);
    int dev = 0;
    (void)hipGetDevice(&dev);
    k.val(dev);
    return cache.run(k, (hipStream_t)stream, direct);
}

// The same with one (B,3,N) gradient pointer per layer and list: autograd hands the node a gradient per output
// tensor (training.py's loss touches ps[0], mus[0] and every logvar), so nothing has to be stacked into (L,B,3,N)
// blocks and the layers whose outputs are unused read nothing.  Any table, and any entry, may be NULL (zero).
extern "C" int dpf_flow_train_backward_lists(int n_layers, int B, int N, int mode, int precision, const int *meta_host,
                                             const float *tcanon, const void *packed, const float *film, const float *stats,
                                             const float *p_in, const float *ps, const float *mus, const float *logvars,
                                             const float *const *g_ps, const float *const *g_mus, const float *const *g_lvs,
                                             float *dp_in, float *dp_tmp, float *dcanon, float *dfm, float flow_eps,
                                             void *workspace, dpf_stream_t stream) {
    const float *const *tab[3] = {g_ps, g_mus, g_lvs};
    auto direct = [&](hipStream_t st) {
        return backward_stack(n_layers, B, N, mode, precision, meta_host, tcanon, packed, film, stats, p_in, ps, mus, logvars,
                              [&](int which, int l) { return tab[which] ? tab[which][l] : nullptr; }, dp_in, dp_tmp, dcanon,
                              dfm, flow_eps, workspace, (dpf_stream_t)st);
    };
    if (n_layers <= 0 || !meta_host) return direct((hipStream_t)stream);
    static GraphCache cache;
    GraphKey k;
    k.val(n_layers); k.val(B); k.val(N); k.val(mode); k.val(precision); k.add(meta_host, sizeof(int) * 4 * n_layers);
    k.val(tcanon); k.val(packed); k.val(film); k.val(stats); k.val(p_in); k.val(ps); k.val(mus); k.val(logvars);
    for (int w = 0; w < 3; ++w) {
        const int present = tab[w] != nullptr;
        k.val(present);
        if (present) k.add(tab[w], sizeof(const float *) * n_layers);      // the per-layer gradient pointers themselves
    }
    k.val(dp_in); k.val(dp_tmp); k.val(dcanon); k.val(dfm); k.val(flow_eps); k.val(workspace);
    int dev = 0;
    (void)hipGetDevice(&dev);
    k.val(dev);
    return cache.run(k, (hipStream_t)stream, direct);
}

// number of training-mode calls served by a graph replay so far in this process (csrc/graph_cache.h); diagnostics / tests
// BatchNorm running statistics of a training step, all 8 L BatchNorm1d layers of the stack in one launch:
//   running = (1 - momentum) * running + momentum * batch      (nn.BatchNorm1d; unbiased batch variance)
// rows [0, 4L): the FiLM nets (batch statistics from dpf_film_train_forward or the caller's tensor ops), rows [4L, 8L): the
// conditioner stacks' BN0 / BN1 of (layer, branch), read from the `stats` block of dpf_flow_train_forward.  The arithmetic is
// the tensor ops' (`mul_` then `add_(batch, alpha=momentum)` = one rounding, then one fused multiply-add): same bits.
namespace {
__global__ __launch_bounds__(256) void trunning_kernel(int n_layers, float keep, float mom, const float *__restrict__ film_mean,
                                                       const float *__restrict__ film_uvar, const float *__restrict__ stats,
                                                       float *__restrict__ rm, float *__restrict__ rv, long long *__restrict__ nbt) {
    const int i = blockIdx.x * 256 + threadIdx.x, r = i >> 6, f = i & 63, nf = 4 * n_layers;
    if (r >= 2 * nf) return;
    float bm, bv;
    if (r < nf) {
        bm = film_mean[i]; bv = film_uvar[i];
    } else {
        const int j = r - nf, l = j >> 2, br = (j >> 1) & 1, which = j & 1;
        const float *st = stats + (size_t)l * ST_LAYER + br * ST_BR;
        bm = st[(which ? 2 : 0) * 64 + f]; bv = st[(which ? 5 : 4) * 64 + f];
    }
    rm[i] = __fmaf_rn(mom, bm, __fmul_rn(rm[i], keep));
    rv[i] = __fmaf_rn(mom, bv, __fmul_rn(rv[i], keep));
    if (f == 0) nbt[r] += 1;
}
}  // namespace

extern "C" int dpf_flow_train_update_running(int n_layers, double momentum, const float *film_mean, const float *film_uvar,
                                             const float *stats, float *running_mean, float *running_var,
                                             long long *num_batches_tracked, dpf_stream_t stream) {
    if (n_layers <= 0 || !film_mean || !film_uvar || !stats || !running_mean || !running_var || !num_batches_tracked) return DPF_EINVAL;
    const int n = 8 * n_layers * 64;
    hipLaunchKernelGGL(trunning_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_layers, (float)(1.0 - momentum),
                       (float)momentum, film_mean, film_uvar, stats, running_mean, running_var, num_batches_tracked);
    return (int)hipGetLastError();
}

// workgroups that gave up waiting for role workgroups of their own launch (pass 1: the column sums of the layer above; pass 2: the
// BN1-backward means) and did the sums themselves -- process-wide; 0 as long as the role workgroups are dispatched first
extern "C" long dpf_train_colsum_fallbacks(void) {
    unsigned v = 0, m = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_colsum_fallbacks), sizeof(v)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&m, HIP_SYMBOL(g_means_fallbacks), sizeof(m)) != hipSuccess) return -1;
    return (long)v + (long)m;
}

extern "C" long dpf_train_graph_replays(void) { return dpf_graph_stats().replays.load(); }
// out[5] = {replays, eager calls, recordings, evictions, uncapturable keys}, process-wide
extern "C" void dpf_train_graph_stats(long *out) {
    if (!out) return;
    GraphStats &g = dpf_graph_stats();
    out[0] = g.replays.load(); out[1] = g.eager.load(); out[2] = g.records.load(); out[3] = g.evictions.load();
    out[4] = g.uncapturable.load();
}
// switch recording / replay on or off at run time (the DPF_TRAIN_GRAPH environment variable sets the initial state);
// returns the previous state.  Recorded graphs are kept while it is off.
extern "C" int dpf_train_graph_set_enabled(int on) { return dpf_graph_enabled_flag().exchange(on ? 1 : 0); }

#ifdef DPF_PROFILE
__global__ void kprof_set_kernel(unsigned long long *p) { g_kprof = p; }
extern "C" void dpf_debug_set_kprof(void *p) {
    hipLaunchKernelGGL(kprof_set_kernel, dim3(1), dim3(1), 0, 0, (unsigned long long *)p);
    hipDeviceSynchronize();
}
extern "C" void dpf_debug_set_tprof(void *p) {
    hipLaunchKernelGGL(tprof_set_kernel, dim3(1), dim3(1), 0, 0, (unsigned long long *)p);
    hipDeviceSynchronize();
}
#endif
