// Latent prior flow, TRAINING mode: GlobalRNVPDecoder.forward under model.train() (BatchNorm1d on the statistics of
// the B rows, lib/networks/flows.py:176-213, decoders.py:21-38) and what autograd derives from it.
//
// The operands are (B x 64..256) matrices: as tensor ops a step is ~25 launches forward and ~50 backward, and the
// 14 steps of a training iteration cost ~12 ms of launch latency -- more than the whole point decoder.  Here a step
// is 4 launches forward and 5 backward, issued back to back by one C call: small fp32 GEMMs through one strided
// LDS-tiled kernel (the kept / warped coordinate sets are strides, not gathers; both nets of a step are one batched
// launch) and four fused element / column kernels (batch statistics + Swish, the affine update, their backwards).
// fp32 FMAs throughout: these matrices are launch-latency-, not throughput-bound, and the tolerance is fp32's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpf_hip.h"

namespace {

constexpr int T = 256;

// floats of one net in the canonical block: W0 | gamma | beta | [running_mean | running_var] | W1 | b1  (nbn = 4 with the
// running-statistics slots of dpf_gprior_pack's layout, 2 in the parameters-only layout)
inline size_t net_floats(int K, int nf, int nbn) { return (size_t)2 * nf * K + (size_t)nbn * nf + K; }

struct Gemm {          // C[m][n] (+)= sum over `nsum` operand pairs, sum over k:  A[m][k] * Bm[k][n]     (blockIdx.z = batch)
    int M, N, Kd, nsum, accumulate;
    const float *A, *Bm;
    float *C;
    long a_rs, a_cs, a_bs, a_ss, b_rs, b_cs, b_bs, b_ss, c_rs, c_cs, c_bs;
};

// 32 x 32 tile per workgroup, inner chunks of KC.  The matrices are (B x 64..256): a launch has 16-64 tiles on 256 CUs, so a
// tile's time is the length of one wave's dependent chain, not throughput -- the four waves of a workgroup each take a
// quarter of every chunk's inner range (4 x 4 outputs per lane, 16 steps of LDS latency per chunk instead of 64) and
// add their partial tiles through LDS at the end.  Lanes fetch along whichever index is contiguous in memory (the
// operands are strided views: weights and their transposes, every other column of the rows, ...).
constexpr int KC = 64;
struct Gemm2 {          // two independent products in one launch: blockIdx.z < batch0 is a batch index of the first, the rest of the second
    Gemm p[2];
    int batch0;
};

__global__ __launch_bounds__(T) void sgemm_kernel(Gemm2 gg) {
    const int prob = (int)blockIdx.z >= gg.batch0;
    const Gemm &g = gg.p[prob];
    if ((int)blockIdx.x * 32 >= g.N || (int)blockIdx.y * 32 >= g.M) return;          // the grid covers the larger of the two
    __shared__ float smem[2 * KC * 36 > 4 * 32 * 33 ? 2 * KC * 36 : 4 * 32 * 33];
    float (*As)[36] = (float (*)[36])smem, (*Bs)[36] = (float (*)[36])(smem + KC * 36);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, tx = lane & 7, ty = lane >> 3;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32, z = blockIdx.z - prob * gg.batch0;
    const bool akf = labs(g.a_cs) < labs(g.a_rs), bkf = labs(g.b_rs) < labs(g.b_cs);
    auto ax = [&](int u) { return akf ? (tid >> 6) + 4 * u : tid & 31; };
    auto ak = [&](int u) { return akf ? tid & 63 : (tid >> 5) + 8 * u; };
    auto bx = [&](int u) { return bkf ? (tid >> 6) + 4 * u : tid & 31; };
    auto bk = [&](int u) { return bkf ? tid & 63 : (tid >> 5) + 8 * u; };
    float acc[4][4] = {};
    float av[8], bv[8];
    const int chunks = (g.Kd + KC - 1) / KC, total = g.nsum * chunks;
    auto fetch = [&](int it) {
        const int si = it / chunks, k0 = (it - si * chunks) * KC;
        const float *A0 = g.A + z * g.a_bs + si * g.a_ss, *B0 = g.Bm + z * g.b_bs + si * g.b_ss;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int xa = m0 + ax(u), ka = k0 + ak(u), xb = n0 + bx(u), kb = k0 + bk(u);
            av[u] = xa < g.M && ka < g.Kd ? A0[(long)xa * g.a_rs + (long)ka * g.a_cs] : 0.f;
            bv[u] = xb < g.N && kb < g.Kd ? B0[(long)kb * g.b_rs + (long)xb * g.b_cs] : 0.f;
        }
    };
    fetch(0);
    for (int it = 0; it < total; ++it) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            As[ak(u)][ax(u)] = av[u];
            Bs[bk(u)][bx(u)] = bv[u];
        }
        __syncthreads();
        if (it + 1 < total) fetch(it + 1);                 // in flight while this chunk is multiplied
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const int k = wave * (KC / 4) + kk;
            const float4 a = *(const float4 *)&As[k][ty * 4], b = *(const float4 *)&Bs[k][tx * 4];
            const float a4[4] = {a.x, a.y, a.z, a.w}, b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
    }
    __syncthreads();
    float (*red)[32][33] = (float (*)[32][33])smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][ty * 4 + i][tx * 4 + j] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * T, mi = e >> 5, ni = e & 31, m = m0 + mi, n = n0 + ni;
        if (m < g.M && n < g.N) {
            const float v = (red[0][mi][ni] + red[1][mi][ni]) + (red[2][mi][ni] + red[3][mi][ni]);
            float *c = g.C + z * g.c_bs + (long)m * g.c_rs + (long)n * g.c_cs;
            *c = g.accumulate ? *c + v : v;
        }
    }
}

void gemm(hipStream_t s, int batch, Gemm g) {
    Gemm2 gg = {{g, g}, batch};
    hipLaunchKernelGGL(sgemm_kernel, dim3((g.N + 31) / 32, (g.M + 31) / 32, batch), dim3(T), 0, s, gg);
}

void gemm2(hipStream_t s, int batch_a, Gemm a, int batch_b, Gemm b) {
    Gemm2 gg = {{a, b}, batch_a};
    const int nx = ((a.N > b.N ? a.N : b.N) + 31) / 32, ny = ((a.M > b.M ? a.M : b.M) + 31) / 32;
    hipLaunchKernelGGL(sgemm_kernel, dim3(nx, ny, batch_a + batch_b), dim3(T), 0, s, gg);
}

__device__ __forceinline__ float swish(float y) { return y / (1.f + expf(-y)); }

// Column kernels: a workgroup owns CW columns; its 256 threads are CW columns x RG row groups, a row group walks
// every RG-th row and the groups' partial sums meet in LDS (a lone thread per column walking all B rows three times was
// a chain of dependent loads: 24 us per launch).
constexpr int CW = 32, RG = T / CW;

__device__ __forceinline__ float column_sum(float v, float (*red)[CW], int cl, int rg) {
    __syncthreads();                     // the previous use of `red`
    red[rg][cl] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) s += red[r][cl];
    return s;
}

// Batch statistics over the B rows of every hidden column (two passes, as torch), BatchNorm affine, Swish.
// h (B, 2 nf) -> stats (2, 2 nf) = mean | biased variance,  hs (B, 2 nf).   cnet = gamma of the step's mu net
__global__ __launch_bounds__(T) void bn_swish_kernel(int B, int nf, size_t cn, float bn_eps, const float *__restrict__ cnet,
                                                     const float *__restrict__ h, float *__restrict__ stats, float *__restrict__ hs) {
    __shared__ float red[RG][CW];
    const int cl = threadIdx.x % CW, rg = threadIdx.x / CW, c = blockIdx.x * CW + cl;
    const bool live = c < 2 * nf;
    const int net = c >= nf, j = c - net * nf;
    float sum = 0.f;
    if (live)
        for (int b = rg; b < B; b += RG) sum += h[(size_t)b * 2 * nf + c];
    const float mean = column_sum(sum, red, cl, rg) / B;
    float sq = 0.f;
    if (live)
        for (int b = rg; b < B; b += RG) {
            const float d = h[(size_t)b * 2 * nf + c] - mean;
            sq = fmaf(d, d, sq);
        }
    const float var = column_sum(sq, red, cl, rg) / B, rstd = 1.f / sqrtf(var + bn_eps);
    if (!live) return;
    if (rg == 0) {
        stats[c] = mean;
        stats[2 * nf + c] = var;
    }
    const float gam = cnet[net * cn + j], bet = cnet[net * cn + nf + j];
    for (int b = rg; b < B; b += RG) hs[(size_t)b * 2 * nf + c] = swish(fmaf((h[(size_t)b * 2 * nf + c] - mean) * rstd, gam, bet));
}

// mu, logvar = log(eps + exp(.)) and the affine update of the warped coordinates; the step's slices of the three lists
__global__ __launch_bounds__(T) void finalize_kernel(int B, int G, int inverse, int kmul, int kadd, int wadd, float eps,
                                                     const float *__restrict__ o, const float *__restrict__ b1m,
                                                     const float *__restrict__ b1l, const float *__restrict__ gin,
                                                     float *__restrict__ gs, float *__restrict__ mus, float *__restrict__ lvs) {
    const int K = G >> 1, e = blockIdx.x * T + threadIdx.x;
    if (e >= B * K) return;
    const int b = e / K, i = e - b * K, wi = kmul * i + wadd, ki = kmul * i + kadd;
    const float om = o[(size_t)b * 2 * K + i] + b1m[i], ol = o[(size_t)b * 2 * K + K + i] + b1l[i];
    const float lv = logf(eps + expf(ol));
    const size_t at = (size_t)b * G;
    const float gw = gin[at + wi];
    gs[at + wi] = inverse ? expf(-0.5f * lv) * (gw - om) : fmaf(expf(0.5f * lv), gw, om);
    gs[at + ki] = gin[at + ki];
    mus[at + wi] = om; mus[at + ki] = 0.f;
    lvs[at + wi] = lv; lvs[at + ki] = 0.f;
}

// Backward of finalize: d_out = dcur + d_gs.  Writes d_o (B, 2K) for the second map, the direct part of the gradient
// w.r.t. the step's input (dnext), and recomputes hs = Swish(BatchNorm(h)) for the weight gradients.
__global__ __launch_bounds__(T) void prep_kernel(int B, int G, int nf, size_t cn, int inverse, int kmul, int kadd, int wadd, float eps,
                                                 float bn_eps, const float *__restrict__ dcur, const float *__restrict__ d_gs,
                                                 const float *__restrict__ d_mus, const float *__restrict__ d_lvs,
                                                 const float *__restrict__ gs, const float *__restrict__ mus,
                                                 const float *__restrict__ lvs, const float *__restrict__ h,
                                                 const float *__restrict__ stats, const float *__restrict__ cnet,
                                                 float *__restrict__ d_o, float *__restrict__ dnext, float *__restrict__ hs) {
    const int K = G >> 1, e = blockIdx.x * T + threadIdx.x;
    if (e < B * K) {
        const int b = e / K, i = e - b * K, wi = kmul * i + wadd, ki = kmul * i + kadd;
        const size_t at = (size_t)b * G;
        const float dw = (dcur ? dcur[at + wi] : 0.f) + (d_gs ? d_gs[at + wi] : 0.f);
        const float dk = (dcur ? dcur[at + ki] : 0.f) + (d_gs ? d_gs[at + ki] : 0.f);
        const float lv = lvs[at + wi], mu = mus[at + wi], out = gs[at + wi];
        const float sc = expf(inverse ? -0.5f * lv : 0.5f * lv);
        const float dmu = (d_mus ? d_mus[at + wi] : 0.f) + (inverse ? -dw * sc : dw);
        const float dlv = (d_lvs ? d_lvs[at + wi] : 0.f) + (inverse ? -0.5f * dw * out : 0.5f * dw * (out - mu));
        d_o[(size_t)b * 2 * K + i] = dmu;
        d_o[(size_t)b * 2 * K + K + i] = dlv * (1.f - eps * expf(-lv));          // d/do log(eps + exp(o)) = exp(o) / (eps + exp(o))
        dnext[at + wi] = dw * sc;
        dnext[at + ki] = dk;
    }
    if (e < B * 2 * nf) {
        const int c = e % (2 * nf), net = c >= nf, j = c - net * nf;
        const float mean = stats[c], rstd = 1.f / sqrtf(stats[2 * nf + c] + bn_eps);
        hs[e] = swish(fmaf((h[e] - mean) * rstd, cnet[net * cn + j], cnet[net * cn + nf + j]));
    }
}

// Backward of Swish and of the batch-statistics BatchNorm, per hidden column; d gamma, d beta; and d b1 = column sums of d_o.
// Workgroups [0, ceil(2nf / CW)) take the hidden columns, the rest the 2K columns of d_o.
__global__ __launch_bounds__(T) void bn_backward_kernel(int B, int G, int nf, size_t cn, int nbn, float bn_eps, const float *__restrict__ cnet,
                                                        const float *__restrict__ h, const float *__restrict__ stats,
                                                        const float *__restrict__ dhs, const float *__restrict__ d_o,
                                                        float *__restrict__ dh, float *__restrict__ dcnet) {
    __shared__ float red[RG][CW];
    const int K = G >> 1, cl = threadIdx.x % CW, rg = threadIdx.x / CW, hidden_blocks = (2 * nf + CW - 1) / CW;
    if ((int)blockIdx.x < hidden_blocks) {
        const int c = blockIdx.x * CW + cl;
        const bool live = c < 2 * nf;
        const int net = c >= nf, j = c - net * nf;
        const float mean = live ? stats[c] : 0.f, rstd = live ? 1.f / sqrtf(stats[2 * nf + c] + bn_eps) : 0.f;
        const float gam = live ? cnet[net * cn + j] : 0.f, bet = live ? cnet[net * cn + nf + j] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        if (live)
            for (int b = rg; b < B; b += RG) {
                const float xh = (h[(size_t)b * 2 * nf + c] - mean) * rstd, y = fmaf(xh, gam, bet), sg = 1.f / (1.f + expf(-y));
                const float dy = dhs[(size_t)b * 2 * nf + c] * (sg * (1.f + y * (1.f - sg)));
                s1 += dy;
                s2 = fmaf(dy, xh, s2);
            }
        s1 = column_sum(s1, red, cl, rg);
        s2 = column_sum(s2, red, cl, rg);
        if (!live) return;
        if (rg == 0) {
            dcnet[net * cn + j] = s2;                  // d gamma
            dcnet[net * cn + nf + j] = s1;             // d beta
            if (nbn == 4) {
                dcnet[net * cn + 2 * nf + j] = 0.f;    // running statistics carry no gradient
                dcnet[net * cn + 3 * nf + j] = 0.f;
            }
        }
        const float m1 = s1 / B, m2 = s2 / B;
        for (int b = rg; b < B; b += RG) {
            const float xh = (h[(size_t)b * 2 * nf + c] - mean) * rstd, y = fmaf(xh, gam, bet), sg = 1.f / (1.f + expf(-y));
            const float dy = dhs[(size_t)b * 2 * nf + c] * (sg * (1.f + y * (1.f - sg)));
            dh[(size_t)b * 2 * nf + c] = gam * rstd * (dy - m1 - xh * m2);
        }
    } else {
        const int c = (blockIdx.x - hidden_blocks) * CW + cl;
        const bool live = c < 2 * K;
        float sm = 0.f;
        if (live)
            for (int b = rg; b < B; b += RG) sm += d_o[(size_t)b * 2 * K + c];
        sm = column_sum(sm, red, cl, rg);
        if (live && rg == 0) {
            const int net = c >= K, i = c - net * K;
            dcnet[net * cn + (size_t)nbn * nf + (size_t)K * nf + i] = sm;      // d b1: behind the BatchNorm vectors and W1 (dcnet starts at gamma)
        }
    }
}

bool steps_ok(int S, int B, int G, int nf, int mode, const int *codes) {
    if (S <= 0 || B < 2 || G < 2 || (G & 1) || nf <= 0 || !codes || (mode != 0 && mode != 1)) return false;
    for (int s = 0; s < S; ++s)
        if (codes[s] < 0 || codes[s] > 3) return false;
    return true;
}

}  // namespace

extern "C" {

size_t dpf_gprior_train_workspace_floats(int B, int G, int n_features) {
    // hs | o / d_o | dhs | dh | two running-gradient buffers
    return (size_t)B * (3 * 2 * (size_t)n_features + G + 2 * (size_t)G);
}

int dpf_gprior_train_forward(int S, int B, int G, int nf, int mode, const int *codes, int params_only, const float *canon, const float *g, float *gs,
                             float *mus, float *lvs, float *save_h, float *save_stats, float *workspace, float bn_eps, float eps,
                             dpf_stream_t stream) {
    if (!steps_ok(S, B, G, nf, mode, codes) || !canon || !g || !gs || !mus || !lvs || !save_h || !save_stats || !workspace) return DPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int K = G / 2;
    const int nbn = params_only ? 2 : 4;
    const size_t cn = net_floats(K, nf, nbn), BG = (size_t)B * G;
    float *hs = workspace, *o = hs + (size_t)B * 2 * nf;
    for (int t = 0; t < S; ++t) {
        const int s = mode ? S - 1 - t : t, code = codes[s];
        const int kmul = code < 2 ? 2 : 1, kadd = code == 0 ? 1 : code == 2 ? K : 0, wadd = code == 1 ? 1 : code == 3 ? K : 0;
        const float *cs = canon + (size_t)s * 2 * cn;
        const float *gin = t == 0 ? g : gs + (size_t)(mode ? s + 1 : s - 1) * BG;
        float *h = save_h + (size_t)s * B * 2 * nf;
        Gemm f1 = {B, nf, K, 1, 0, gin + kadd, cs, h, G, kmul, 0, 0, 1, K, (long)cn, 0, 2L * nf, 1, nf};
        gemm(st, 2, f1);
        hipLaunchKernelGGL(bn_swish_kernel, dim3((2 * nf + CW - 1) / CW), dim3(T), 0, st, B, nf, cn, bn_eps, cs + (size_t)nf * K, h,
                           save_stats + (size_t)s * 4 * nf, hs);
        const float *w1 = cs + (size_t)nf * K + (size_t)nbn * nf;
        Gemm f2 = {B, K, nf, 1, 0, hs, w1, o, 2L * nf, 1, nf, 0, 1, nf, (long)cn, 0, 2L * K, 1, K};
        gemm(st, 2, f2);
        hipLaunchKernelGGL(finalize_kernel, dim3((B * K + T - 1) / T), dim3(T), 0, st, B, G, mode, kmul, kadd, wadd, eps, o,
                           w1 + (size_t)K * nf, w1 + (size_t)K * nf + cn, gin, gs + (size_t)s * BG, mus + (size_t)s * BG,
                           lvs + (size_t)s * BG);
    }
    return (int)hipGetLastError();
}

int dpf_gprior_train_backward(int S, int B, int G, int nf, int mode, const int *codes, int params_only, const float *canon, const float *g,
                              const float *gs, const float *mus, const float *lvs, const float *save_h, const float *save_stats,
                              const float *d_gs, const float *d_mus, const float *d_lvs, float *dg, float *dcanon, float *workspace,
                              float bn_eps, float eps, dpf_stream_t stream) {
    if (!steps_ok(S, B, G, nf, mode, codes) || !canon || !g || !gs || !mus || !lvs || !save_h || !save_stats || !dg || !dcanon || !workspace)
        return DPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int K = G / 2;
    const int nbn = params_only ? 2 : 4;
    const size_t cn = net_floats(K, nf, nbn), BG = (size_t)B * G, BH = (size_t)B * 2 * nf;
    float *hs = workspace, *d_o = hs + BH, *dhs = d_o + BG, *dh = dhs + BH, *run[2] = {dh + BH, dh + BH + BG};
    const float *dcur = nullptr;
    for (int t = S - 1; t >= 0; --t) {                       // the forward's steps, last one first
        const int s = mode ? S - 1 - t : t, code = codes[s];
        const int kmul = code < 2 ? 2 : 1, kadd = code == 0 ? 1 : code == 2 ? K : 0, wadd = code == 1 ? 1 : code == 3 ? K : 0;
        const float *cs = canon + (size_t)s * 2 * cn;
        float *dcs = dcanon + (size_t)s * 2 * cn;
        const float *gin = t == 0 ? g : gs + (size_t)(mode ? s + 1 : s - 1) * BG;
        const float *h = save_h + (size_t)s * BH, *stats = save_stats + (size_t)s * 4 * nf;
        float *dnext = t == 0 ? dg : run[t & 1];
        const int n_el = B * K > B * 2 * nf ? B * K : B * 2 * nf;
        hipLaunchKernelGGL(prep_kernel, dim3((n_el + T - 1) / T), dim3(T), 0, st, B, G, nf, cn, mode, kmul, kadd, wadd, eps, bn_eps, dcur,
                           d_gs ? d_gs + (size_t)s * BG : nullptr, d_mus ? d_mus + (size_t)s * BG : nullptr,
                           d_lvs ? d_lvs + (size_t)s * BG : nullptr, gs + (size_t)s * BG, mus + (size_t)s * BG, lvs + (size_t)s * BG, h, stats,
                           cs + (size_t)nf * K, d_o, dnext, hs);
        const float *w1 = cs + (size_t)nf * K + (size_t)nbn * nf;
        float *dw1 = dcs + (size_t)nf * K + (size_t)nbn * nf;
        Gemm g1 = {B, nf, K, 1, 0, d_o, w1, dhs, 2L * K, 1, K, 0, nf, 1, (long)cn, 0, 2L * nf, 1, nf};            // d hs = d_o W1
        Gemm g2 = {K, nf, B, 1, 0, d_o, hs, dw1, 1, 2L * K, K, 0, 2L * nf, 1, nf, 0, nf, 1, (long)cn};            // d W1 = d_o^T hs
        gemm2(st, 2, g1, 2, g2);                                                                                      // independent: one launch
        hipLaunchKernelGGL(bn_backward_kernel, dim3((2 * nf + CW - 1) / CW + (2 * K + CW - 1) / CW), dim3(T), 0, st, B, G, nf, cn, nbn, bn_eps, cs + (size_t)nf * K, h, stats,
                           dhs, d_o, dh, dcs + (size_t)nf * K);
        Gemm g3 = {B, K, nf, 2, 1, dh, cs, dnext + kadd, 2L * nf, 1, 0, nf, K, 1, 0, (long)cn, G, kmul, 0};        // d g_keep += sum_net dh W0
        Gemm g4 = {nf, K, B, 1, 0, dh, gin + kadd, dcs, 1, 2L * nf, nf, 0, G, kmul, 0, 0, K, 1, (long)cn};        // d W0 = dh^T g_keep
        gemm2(st, 1, g3, 2, g4);                                                                                   // independent: one launch
        dcur = dnext;
    }
    return (int)hipGetLastError();
}

}  // extern "C"
