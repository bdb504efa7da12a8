// Training-mode FiLM conditioner nets of the coupling stack, all K = 4 L of them in ONE launch each way (gfx950).
//
// Replaces, for model.train(), what lib/networks/flows.py:33-45, 68-80 builds per sub-net and autograd derives from it:
//     u = g W0^T            Linear(G, F, bias=False)                      (B, F)
//     xhat = BatchNorm1d(u) batch statistics over the B clouds, biased variance, eps 1e-5
//     y = gamma * xhat + beta ;  sw = y * sigmoid(y)                       Swish (layers.py:9-10)
//     fm = sw W1^T + b1     Linear(F, F)                                   (B, F)  -> the stack's `fm` input
// SURVEY 8(a4) leaves these nets on PyTorch-ROCm ("negligible FLOPs but 4 x 4 tiny kernels per layer x 63"): batched over
// the K nets they were ~12 tensor-op launches forward and ~25 backward per optimizer step -- 0.35 ms of the 4.46 ms step
// (profiles/r03_train_kernel_trace.txt: ~55 launches of 5-8 us between the stack's two graphs).  Here: one workgroup per
// sub-net, fp32 VALU (every product exact fp32 as in the tensor ops; sums in a fixed order), activations in registers / LDS.
//
// Threads: 256 = 64 features x 4 cloud groups; thread (f, q) owns feature f of clouds q, q + 4, ...  (NB of them, B <= 4 NB).
// Forward saves xhat (K, B, F) and rstd (K, F); the backward recomputes y, sigmoid, swish from them.
#include <stdint.h>

#include "../../include/dpf_hip.h"
#include "lds_attr.h"

namespace {

constexpr int F = 64;
constexpr int T = 256;
constexpr int GT = 128;          // columns of g staged per tile

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FwdArgs {
    int B, G;
    float eps;
    const float *g;              // (B, G)
    const float *W0;             // (K, F, G)
    const float *gam, *bet;      // (K, F)
    const float *W1;             // (K, F, F)   [out][in]
    const float *b1;             // (K, F)
    float *fm;                   // (K, B, F)
    float *xhat;                 // (K, B, F)
    float *rstd, *mean, *uvar;   // (K, F): 1/sqrt(var + eps), batch mean, unbiased batch variance (running_var update)
};

__device__ __forceinline__ float sigmoidf_(float y) { return 1.0f / (1.0f + expf(-y)); }

// sum over the four cloud groups of a per-thread partial, in group order (red: [4][64] floats of LDS)
__device__ __forceinline__ float group_sum(float v, float *red, int f, int q) {
    __syncthreads();                              // the previous use of `red`
    red[q * F + f] = v;
    __syncthreads();
    return ((red[f] + red[F + f]) + red[2 * F + f]) + red[3 * F + f];
}

constexpr int WP = GT + 4;       // row pitch of a staged 64 x GT weight tile: 16-byte aligned rows, conflict-free for lane = row
constexpr int W1P = F + 4;       // the same for the 64 x 64 matrix

// a (64, ncols <= GT) tile of a row-major matrix (leading dimension ld) <-> LDS [64][WP]: 16-byte accesses, lanes along a row
__device__ __forceinline__ void tile_load(float *lds, const float *src, size_t ld, int ncols, int pitch, int tid) {
    const int per_row = ncols / 4;
    for (int e = tid; e < F * per_row; e += T) {
        const int r = e / per_row, c = (e % per_row) * 4;
        *(f32x4 *)(lds + r * pitch + c) = *(const f32x4 *)(src + (size_t)r * ld + c);
    }
}

template <int NB>
__global__ __launch_bounds__(T) void film_train_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = a.B, G = a.G, k = blockIdx.x;
    float *gs = sm;                               // [B][GT]
    float *us = sm + (size_t)B * GT;              // [B][F]  (swish(y))
    float *red = us + (size_t)B * F;              // [4][F]
    float *ws = red + 4 * F;                      // [F][WP]  W0 tile, later W1 [F][W1P]
    const int tid = threadIdx.x, f = tid & 63, q = tid >> 6;
    float acc[NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) acc[bi] = 0.f;
    for (int gt = 0; gt < G; gt += GT) {
        const int gw = G - gt < GT ? G - gt : GT;                      // G % 4 == 0 (checked by the launcher)
        __syncthreads();                                               // the previous tile has been consumed
        for (int e = tid; e < B * (GT / 4); e += T) {
            const int b = e / (GT / 4), c = (e % (GT / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < gw) v = *(const f32x4 *)(a.g + (size_t)b * G + gt + c);
            *(f32x4 *)(gs + (size_t)b * GT + c) = v;
        }
        // the sub-net's W0 rows of this tile: coalesced loads, all in flight (read per lane = row straight from global
        // memory inside the loop below, every iteration waited for an uncoalesced load: 25 us per launch)
        tile_load(ws, a.W0 + (size_t)k * F * G + gt, G, gw, WP, tid);
        __syncthreads();
        for (int c = 0; c < gw; c += 4) {
            const f32x4 w = *(const f32x4 *)(ws + f * WP + c);
            // (a cloud index past the batch is clamped, its sum never used: `if (b < B)` around the body became eight
            // exec-masked blocks per step, each waiting for its own LDS read -- 16 of the launch's 25 us)
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                const int b = q + 4 * bi < B ? q + 4 * bi : B - 1;
                const f32x4 x = *(const f32x4 *)(gs + (size_t)b * GT + c);              // same address in the whole wave: broadcast
                acc[bi] = __builtin_fmaf(w.w, x.w, __builtin_fmaf(w.z, x.z, __builtin_fmaf(w.y, x.y, __builtin_fmaf(w.x, x.x, acc[bi]))));
            }
        }
    }
    // ---- BatchNorm1d over the batch dimension (two passes, biased variance)
    float s = 0.f;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) s += q + 4 * bi < B ? acc[bi] : 0.f;
    const float mean = group_sum(s, red, f, q) / (float)B;
    float v = 0.f;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const float d = acc[bi] - mean;
        v += q + 4 * bi < B ? d * d : 0.f;
    }
    const float var = group_sum(v, red, f, q) / (float)B;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (q == 0) {
        a.rstd[(size_t)k * F + f] = rstd;
        a.mean[(size_t)k * F + f] = mean;
        a.uvar[(size_t)k * F + f] = var * ((float)B / (float)(B - 1));
    }
    const float gam = a.gam[(size_t)k * F + f], bet = a.bet[(size_t)k * F + f];
    tile_load(ws, a.W1 + (size_t)k * F * F, F, F, W1P, tid);           // (every thread is past the W0 tile: group_sum's barriers)
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int b = q + 4 * bi;
        if (b < B) {
            const float xh = (acc[bi] - mean) * rstd;
            const float y = __builtin_fmaf(xh, gam, bet);
            a.xhat[((size_t)k * B + b) * F + f] = xh;
            us[(size_t)b * F + f] = y * sigmoidf_(y);
        }
    }
    __syncthreads();
    // ---- Linear(F, F) + bias: thread (f, q) = output feature f of its clouds; its W1 row in registers
    f32x4 w1[F / 4];
#pragma unroll
    for (int i = 0; i < F / 4; ++i) w1[i] = *(const f32x4 *)(ws + f * W1P + 4 * i);
    const float b1 = a.b1[(size_t)k * F + f];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int b = q + 4 * bi, bc = b < B ? b : B - 1;
        float o = b1;
#pragma unroll
        for (int i = 0; i < F / 4; ++i) {
            const f32x4 x = *(const f32x4 *)(us + (size_t)bc * F + 4 * i);
            o = __builtin_fmaf(w1[i].w, x.w, __builtin_fmaf(w1[i].z, x.z, __builtin_fmaf(w1[i].y, x.y, __builtin_fmaf(w1[i].x, x.x, o))));
        }
        if (b < B) a.fm[((size_t)k * B + b) * F + f] = o;
    }
}

struct BwdArgs {
    int B, G, accumulate;
    const float *g;              // (B, G)
    const float *W0, *gam, *bet, *W1;
    const float *xhat;           // (K, B, F)
    const float *rstd;           // (K, F)
    const float *dfm;            // (K, B, F)   d loss / d fm
    float *dW0;                  // (K, F, G)
    float *dgam, *dbet;          // (K, F)
    float *dW1;                  // (K, F, F)
    float *db1;                  // (K, F)
    float *dg_part;              // (K, B, G) or NULL: this sub-net's share of d loss / d g
};

__device__ __forceinline__ void put(float *p, float v, int accumulate) { *p = accumulate ? *p + v : v; }
// LDS [64][pitch] -> a (64, ncols) tile of a row-major matrix, overwritten or added to: 16-byte accesses, lanes along a row
__device__ __forceinline__ void tile_store(float *dst, size_t ld, const float *lds, int ncols, int pitch, int tid, int accumulate) {
    const int per_row = ncols / 4;
    for (int e = tid; e < F * per_row; e += T) {
        const int r = e / per_row, c = (e % per_row) * 4;
        f32x4 v = *(const f32x4 *)(lds + r * pitch + c);
        f32x4 *o = (f32x4 *)(dst + (size_t)r * ld + c);
        if (accumulate) { const f32x4 old = *o; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
        *o = v;
    }
}

template <int NB>
__global__ __launch_bounds__(T) void film_train_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = a.B, G = a.G, k = blockIdx.x;
    float *d0 = sm;                               // [B][F]  d fm, later d u
    float *sws = d0 + (size_t)B * F;              // [B][F]  swish(y)
    float *w1s = sws + (size_t)B * F;             // [F][F + 1]
    float *red = w1s + F * (F + 1);               // [4][F]
    float *gs = red + 4 * F;                      // [B][GT]
    const int tid = threadIdx.x, f = tid & 63, q = tid >> 6;
    const float gam = a.gam[(size_t)k * F + f], bet = a.bet[(size_t)k * F + f], rstd = a.rstd[(size_t)k * F + f];
    // ---- stage d fm and W1, recompute y / sigmoid / swish of this thread's (cloud, feature) entries
    for (int e = tid; e < B * (F / 4); e += T)
        *(f32x4 *)(d0 + 4 * (size_t)e) = *(const f32x4 *)(a.dfm + (size_t)k * B * F + 4 * (size_t)e);
    for (int e = tid; e < F * (F / 4); e += T) {
        const int r = e / (F / 4), c = (e % (F / 4)) * 4;
        const f32x4 w = *(const f32x4 *)(a.W1 + ((size_t)k * F + r) * F + c);
        w1s[r * (F + 1) + c] = w.x; w1s[r * (F + 1) + c + 1] = w.y; w1s[r * (F + 1) + c + 2] = w.z; w1s[r * (F + 1) + c + 3] = w.w;
    }
    float xh[NB], yv[NB], sg[NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        const int b = q + 4 * bi;
        xh[bi] = 0.f; yv[bi] = 0.f; sg[bi] = 0.f;
        if (b < B) {
            xh[bi] = a.xhat[((size_t)k * B + b) * F + f];
            yv[bi] = __builtin_fmaf(xh[bi], gam, bet);
            sg[bi] = sigmoidf_(yv[bi]);
            sws[(size_t)b * F + f] = yv[bi] * sg[bi];
        }
    }
    __syncthreads();
    // ---- d b1 = sum_b d fm
    {
        float s = 0.f;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) s += q + 4 * bi < B ? d0[(size_t)(q + 4 * bi) * F + f] : 0.f;
        const float t = group_sum(s, red, f, q);
        if (q == 0) put(a.db1 + (size_t)k * F + f, t, a.accumulate);
    }
    // ---- d W1[f'][c] = sum_b d fm[b][f'] * sw[b][c]:  thread (f' = f, q) owns the 16 columns c = 16 q ..
    {
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = d0[(size_t)b * F + f];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 x = *(const f32x4 *)(sws + (size_t)b * F + 16 * q + 4 * i);
                w[4 * i + 0] = __builtin_fmaf(d, x.x, w[4 * i + 0]); w[4 * i + 1] = __builtin_fmaf(d, x.y, w[4 * i + 1]);
                w[4 * i + 2] = __builtin_fmaf(d, x.z, w[4 * i + 2]); w[4 * i + 3] = __builtin_fmaf(d, x.w, w[4 * i + 3]);
            }
        }
        // through LDS to the output as 16-byte read-modify-writes with lanes along a row (gs is free until the d W0 pass)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4 *)(gs + f * W1P + 16 * q + 4 * i) = f32x4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
        __syncthreads();
        tile_store(a.dW1 + (size_t)k * F * F, F, gs, F, W1P, tid, a.accumulate);
    }
    // ---- d sw = d fm W1, back through Swish and the BatchNorm
    float du[NB];
    {
        float w1c[F];                                                  // column f of W1
#pragma unroll
        for (int r = 0; r < F; ++r) w1c[r] = w1s[r * (F + 1) + f];
        float pg = 0.f, pb = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            const int b = q + 4 * bi, bc = b < B ? b : B - 1;          // (clamped, branch-free: see the forward kernel)
            float ds = 0.f;
#pragma unroll
            for (int i = 0; i < F / 4; ++i) {
                const f32x4 x = *(const f32x4 *)(d0 + (size_t)bc * F + 4 * i);
                ds = __builtin_fmaf(x.w, w1c[4 * i + 3], __builtin_fmaf(x.z, w1c[4 * i + 2], __builtin_fmaf(x.y, w1c[4 * i + 1], __builtin_fmaf(x.x, w1c[4 * i], ds))));
            }
            const float dy = b < B ? ds * (sg[bi] * (1.0f + yv[bi] * (1.0f - sg[bi]))) : 0.f;      // xh = y = sigmoid = 0 past the batch anyway
            pg = __builtin_fmaf(dy, xh[bi], pg);
            pb += dy;
            const float dxh = dy * gam;
            p1 += dxh;
            p2 = __builtin_fmaf(dxh, xh[bi], p2);
            du[bi] = dxh;
        }
        const float tg = group_sum(pg, red, f, q);
        const float tb = group_sum(pb, red, f, q);
        const float m1 = group_sum(p1, red, f, q) / (float)B;
        const float m2 = group_sum(p2, red, f, q) / (float)B;
        if (q == 0) {
            put(a.dgam + (size_t)k * F + f, tg, a.accumulate);
            put(a.dbet + (size_t)k * F + f, tb, a.accumulate);
        }
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) du[bi] = rstd * (du[bi] - m1 - xh[bi] * m2);
    }
    __syncthreads();                                                   // everyone is done with d fm
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
        if (q + 4 * bi < B) d0[(size_t)(q + 4 * bi) * F + f] = du[bi];
    // ---- d W0[f][j] = sum_b d u[b][f] * g[b][j]:  tile of GT columns, thread (f, q) owns GT / 4 of them
    for (int gt = 0; gt < G; gt += GT) {
        const int gw = G - gt < GT ? G - gt : GT;
        __syncthreads();                                               // d u published / the previous tile consumed
        for (int e = tid; e < B * (GT / 4); e += T) {
            const int b = e / (GT / 4), c = (e % (GT / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < gw) v = *(const f32x4 *)(a.g + (size_t)b * G + gt + c);
            *(f32x4 *)(gs + (size_t)b * GT + c) = v;
        }
        __syncthreads();
        float w[GT / 4];
#pragma unroll
        for (int i = 0; i < GT / 4; ++i) w[i] = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = d0[(size_t)b * F + f];
#pragma unroll
            for (int i = 0; i < GT / 16; ++i) {
                const f32x4 x = *(const f32x4 *)(gs + (size_t)b * GT + (GT / 4) * q + 4 * i);
                w[4 * i + 0] = __builtin_fmaf(d, x.x, w[4 * i + 0]); w[4 * i + 1] = __builtin_fmaf(d, x.y, w[4 * i + 1]);
                w[4 * i + 2] = __builtin_fmaf(d, x.z, w[4 * i + 2]); w[4 * i + 3] = __builtin_fmaf(d, x.w, w[4 * i + 3]);
            }
        }
        __syncthreads();                                               // everyone is done reading the g tile
#pragma unroll
        for (int i = 0; i < GT / 16; ++i)
            *(f32x4 *)(gs + f * WP + (GT / 4) * q + 4 * i) = f32x4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
        __syncthreads();
        tile_store(a.dW0 + (size_t)k * F * G + gt, G, gs, gw, WP, tid, a.accumulate);
    }
    // ---- this sub-net's share of d g[b][j] = sum_f d u[b][f] * W0[f][j]:  64 columns at a time, thread (j, q)
    if (a.dg_part != nullptr) {
        for (int j0 = 0; j0 < G; j0 += 64) {
            const int j = j0 + f;
            float wc[F];
#pragma unroll
            for (int r = 0; r < F; ++r) wc[r] = j < G ? a.W0[((size_t)k * F + r) * G + j] : 0.f;      // coalesced over j
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                const int b = q + 4 * bi, bc = b < B ? b : B - 1;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < F / 4; ++i) {
                    const f32x4 x = *(const f32x4 *)(d0 + (size_t)bc * F + 4 * i);
                    s = __builtin_fmaf(x.w, wc[4 * i + 3], __builtin_fmaf(x.z, wc[4 * i + 2], __builtin_fmaf(x.y, wc[4 * i + 1], __builtin_fmaf(x.x, wc[4 * i], s))));
                }
                if (b < B && j < G) a.dg_part[((size_t)k * B + b) * G + j] = s;
            }
        }
    }
}

size_t fwd_lds(int B) { return ((size_t)B * GT + (size_t)B * F + 4 * F + (size_t)F * WP) * sizeof(float); }
size_t bwd_lds(int B) {
    const size_t tile = (size_t)B * GT > (size_t)F * WP ? (size_t)B * GT : (size_t)F * WP;      // the g tile, also the staging tile of dW0 / dW1
    return (2 * (size_t)B * F + F * (F + 1) + 4 * F + tile) * sizeof(float);
}

}  // namespace

// B <= 64: at 4 x 32 clouds per thread the backward kernel's per-cloud registers spill (bigger batches stay on the tensor ops)
extern "C" int dpf_film_train_max_batch(void) { return 64; }

extern "C" int dpf_film_train_forward(int K, int B, int G, const float *g, const float *W0, const float *gamma, const float *beta,
                                      const float *W1, const float *b1, float bn_eps, float *fm, float *xhat, float *rstd,
                                      float *mean, float *uvar, dpf_stream_t stream) {
    if (K <= 0 || B < 2 || G <= 0 || !g || !W0 || !gamma || !beta || !W1 || !b1 || !fm || !xhat || !rstd || !mean || !uvar) return DPF_EINVAL;
    if (B > 64 || (G & 3)) return DPF_ENOSUP;
    FwdArgs a = {B, G, bn_eps, g, W0, gamma, beta, W1, b1, fm, xhat, rstd, mean, uvar};
    const int lds = (int)fwd_lds(B);
    hipStream_t s = (hipStream_t)stream;
#define DPF_FWD(NB)                                                                                         \
    {                                                                                                       \
        static LdsLimit lim;                                                                                \
        if (hipError_t e = lim.ensure((const void *)film_train_fwd_kernel<NB>, lds); e != hipSuccess) return (int)e; \
        hipLaunchKernelGGL(film_train_fwd_kernel<NB>, dim3(K), dim3(T), lds, s, a);                         \
    }
    if (B <= 16) DPF_FWD(4) else if (B <= 32) DPF_FWD(8) else DPF_FWD(16)
#undef DPF_FWD
    return (int)hipGetLastError();
}

extern "C" int dpf_film_train_backward(int K, int B, int G, const float *g, const float *W0, const float *gamma, const float *beta,
                                       const float *W1, const float *xhat, const float *rstd, const float *dfm, float *dW0,
                                       float *dgamma, float *dbeta, float *dW1, float *db1, float *dg_part, int accumulate,
                                       dpf_stream_t stream) {
    if (K <= 0 || B < 2 || G <= 0 || !g || !W0 || !gamma || !beta || !W1 || !xhat || !rstd || !dfm || !dW0 || !dgamma || !dbeta ||
        !dW1 || !db1)
        return DPF_EINVAL;
    if (B > 64 || (G & 3)) return DPF_ENOSUP;
    BwdArgs a = {B, G, accumulate, g, W0, gamma, beta, W1, xhat, rstd, dfm, dW0, dgamma, dbeta, dW1, db1, dg_part};
    const int lds = (int)bwd_lds(B);
    hipStream_t s = (hipStream_t)stream;
#define DPF_BWDK(NB)                                                                                        \
    {                                                                                                       \
        static LdsLimit lim;                                                                                \
        if (hipError_t e = lim.ensure((const void *)film_train_bwd_kernel<NB>, lds); e != hipSuccess) return (int)e; \
        hipLaunchKernelGGL(film_train_bwd_kernel<NB>, dim3(K), dim3(T), lds, s, a);                         \
    }
    if (B <= 16) DPF_BWDK(4) else if (B <= 32) DPF_BWDK(8) else DPF_BWDK(16)
#undef DPF_BWDK
    return (int)hipGetLastError();
}
