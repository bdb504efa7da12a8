// Latent prior flow: the whole GlobalRNVPDecoder stack on (B, G) codes in ONE launch (eval-mode BatchNorm).
//
// Replaces GlobalRNVPDecoder.forward (lib/networks/decoders.py:21-38) = 2*n_flows RealNVPFlow steps
// (lib/networks/flows.py:198-213), each  Linear(K -> nf, no bias) . BatchNorm1d . Swish . Linear(nf -> K)  twice
// (mu and logvar nets) on the K = G/2 kept coordinates, then an affine update of the other K coordinates.
// As tensor ops that is ~25 launches per step on (B, 64..256) operands -- 350+ dependent launches per call,
// 2-3 ms of launch latency around 60 MFLOP.  Here a workgroup owns RB rows of the batch and walks all steps: the
// rows live in LDS, the weights (0.13-0.5 MB per step) stream from L2, nothing but the result lists touches HBM.
// fp32 FMAs in k order -- the work is latency- not throughput-bound, matrix cores would buy nothing.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpf_hip.h"

namespace {

constexpr int THREADS = 1024;      // the steps are chains of dependent round trips: many short dot products, not few long ones
constexpr int PACK_THREADS = 256;
constexpr int MAX_STEPS = 256;

// step code (2 bits): which coordinates a step warps, RealNVPFlowCouple's two patterns (flows.py:224-233)
//   0: even (keep odd)   1: odd (keep even)   2: first half (keep second)   3: second half (keep first)
__host__ __device__ inline size_t canon_net_floats(int K, int nf) { return (size_t)2 * nf * K + 4 * (size_t)nf + K; }
__host__ __device__ inline size_t packed_net_floats(int K, int nf) { return (size_t)2 * nf * K + 2 * (size_t)nf + K; }

// canon (per step, per net: mu then logvar; the reference's state_dict order, flows.py:176-196):
//   W0 [nf][K] | bn.weight | bn.bias | bn.running_mean | bn.running_var | W1 [K][nf] | b1 [K]
// packed (per step, per net):  W0t [K][nf] | a [nf] | c [nf] | W1t [nf][K] | b1 [K]      a = gamma / sqrt(var + eps), c = beta - mean a
__global__ __launch_bounds__(PACK_THREADS) void gprior_pack_kernel(int nets, int K, int nf, float bn_eps, const float *__restrict__ canon,
                                                              float *__restrict__ packed) {
    const size_t cn = canon_net_floats(K, nf), pn = packed_net_floats(K, nf);
    const size_t total = (size_t)nets * pn;
    for (size_t e = (size_t)blockIdx.x * PACK_THREADS + threadIdx.x; e < total; e += (size_t)gridDim.x * PACK_THREADS) {
        const size_t net = e / pn;
        size_t o = e - net * pn;
        const float *c = canon + net * cn;
        const float *gam = c + (size_t)nf * K, *bet = gam + nf, *rm = bet + nf, *rv = rm + nf, *w1 = rv + nf, *b1 = w1 + (size_t)K * nf;
        float v;
        if (o < (size_t)K * nf) {
            const int k = (int)(o / nf), j = (int)(o - (size_t)k * nf);
            v = c[(size_t)j * K + k];
        } else if ((o -= (size_t)K * nf) < (size_t)nf) {
            v = gam[o] / sqrtf(rv[o] + bn_eps);
        } else if ((o -= nf) < (size_t)nf) {
            const float a = gam[o] / sqrtf(rv[o] + bn_eps);
            v = bet[o] - rm[o] * a;
        } else if ((o -= nf) < (size_t)nf * K) {
            const int j = (int)(o / K), i = (int)(o - (size_t)j * K);
            v = w1[(size_t)i * nf + j];
        } else {
            v = b1[o - (size_t)nf * K];
        }
        packed[e] = v;
    }
}

struct GArgs {
    int S, B, G, nf, inverse, parts1, parts2;
    float eps;
    const float *packed, *g;
    float *gs, *mus, *lvs, *sum_lv, *g_out;
    uint32_t codes[MAX_STEPS / 16];
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store to be
// acknowledged, and each step stores its slice of the result lists -- nobody in the kernel reads those back.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}

// acc += w * x (x one number), component by component.  r06: written as a vector expression this became "v_pk_fma_f32 acc, w, X
// op_sel:[0,1,0]" -- both halves multiplied by the HIGH word of a register pair that holds two consecutive x -- and that is the
// one packed form that gfx950 gets wrong: a packed fp32 instruction whose LOW half reads the HIGH word of a VGPR source loses that
// half in lanes 48-63 while another wave of the SIMD issues MFMAs at certain distances (tools/ubench/pk_vs_mfma_forms.hip,
// profiles/r06_packed_f32_vs_mfma.txt).  This kernel issues no MFMA, but a kernel of another stream may (tests/diag/interference_probe.py).
template <int V, class Vec>
__device__ __forceinline__ void fma_splat(Vec &acc, const Vec &w, float x) {
#pragma unroll
    for (int c = 0; c < V; ++c) acc[c] = __builtin_fmaf(w[c], x, acc[c]);
}

// One map phase: out[p][r][q .. q+V) = sum over the p-th part of the inner range of  w[inner][q .. q+V) * x[r][inner],
// q over `width` outputs (V consecutive ones per thread: one 4 V-byte load per V FMAs), x read through a linear
// index map (xmul * inner + xadd: the kept coordinates of the row, or the hidden activations).
template <int RB, int V>
__device__ __forceinline__ void map_phase(const float *__restrict__ w, int width, int inner, int parts, const float *x, int xstride,
                                          int xmul, int xadd, int netsplit, int xnet, size_t wnet, float *part, int tid) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const int nq = width / V;
    for (int o = tid; o < parts * nq; o += THREADS) {
        const int p = o / nq, q = (o - p * nq) * V, net = q >= netsplit;
        const int i0 = inner * p / parts, i1 = inner * (p + 1) / parts;
        const float *wp = w + net * wnet + (q - net * netsplit);
        const float *xp = x + net * xnet + xadd;
        vec acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = (vec)(0.f);
        int k = i0;
        constexpr int U = V == 1 ? 16 : 4;
        for (; k + U <= i1; k += U) {
            vec wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) wv[u] = *(const vec *)(wp + (size_t)(k + u) * netsplit);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < RB; ++r) fma_splat<V>(acc[r], wv[u], xp[r * xstride + (k + u) * xmul]);
        }
        for (; k < i1; ++k) {
            const vec wv = *(const vec *)(wp + (size_t)k * netsplit);
#pragma unroll
            for (int r = 0; r < RB; ++r) fma_splat<V>(acc[r], wv, xp[r * xstride + k * xmul]);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) *(vec *)(part + (size_t)(p * RB + r) * width + q) = acc[r];
    }
}

template <int RB, int V>
__global__ __launch_bounds__(THREADS) void gprior_kernel(GArgs a) {
    extern __shared__ float lds[];
    const int G = a.G, K = G >> 1, nf = a.nf, P1 = a.parts1, P = a.parts2;
    float *gcur = lds;                          // [RB][G]   the rows as they stand
    float *tot = gcur + RB * G;                 // [RB][G]   running sum of the logvars
    float *hs = tot + RB * G;                   // [RB][2 nf] hidden activations, mu net then logvar net
    float *part = hs + RB * 2 * nf;             // [P1][RB][2 nf] partial first-map sums, then [P][RB][2 K] partial second-map sums
    const int tid = threadIdx.x, row0 = blockIdx.x * RB;
    for (int e = tid; e < RB * G; e += THREADS) {
        const int r = e / G, row = row0 + r;
        gcur[e] = row < a.B ? a.g[(size_t)row * G + (e - r * G)] : 0.f;
        tot[e] = 0.f;
    }
    lds_barrier();
    const size_t pn = packed_net_floats(K, nf);
    for (int t = 0; t < a.S; ++t) {
        const int s = a.inverse ? a.S - 1 - t : t;
        const int code = (a.codes[s >> 4] >> ((s & 15) * 2)) & 3;
        // kept coordinate k of the step sits at kmul * k + kadd, warped coordinate i at kmul * i + wadd
        const int kmul = code < 2 ? 2 : 1, kadd = code == 0 ? 1 : code == 2 ? K : 0, wadd = code == 1 ? 1 : code == 3 ? K : 0;
        const float *pk = a.packed + (size_t)s * 2 * pn;
        // the step's BatchNorm vectors and biases are fetched now, with the first-map weights, not when they are needed:
        // a step is a chain of dependent round trips to L2 / the infinity cache, and these two would be links of it
        float bn_a = 0.f, bn_c = 0.f, b_mu = 0.f, b_lv = 0.f;
        if (tid < RB * 2 * nf) {
            const int q = tid % (2 * nf), net = q >= nf, j = q - net * nf;
            bn_a = pk[net * pn + (size_t)K * nf + j];
            bn_c = pk[net * pn + (size_t)K * nf + nf + j];
        }
        if (tid < RB * K) {
            const int i = tid % K;
            b_mu = pk[(size_t)2 * K * nf + 2 * nf + i];
            b_lv = pk[pn + (size_t)2 * K * nf + 2 * nf + i];
        }
        // ---- first map: (part of the kept range, net, j): W0t [K][nf] against the kept coordinates
        map_phase<RB, V>(pk, 2 * nf, K, P1, gcur, G, kmul, kadd, nf, 0, pn, part, tid);
        lds_barrier();
        // ---- BatchNorm (folded) + Swish
        for (int o = tid; o < RB * 2 * nf; o += THREADS) {
            const int r = o / (2 * nf), q = o - r * 2 * nf, net = q >= nf, j = q - net * nf;
            float acc = 0.f;
            for (int p = 0; p < P1; ++p) acc += part[(p * RB + r) * 2 * nf + q];
            const float y = o < THREADS ? fmaf(acc, bn_a, bn_c) : fmaf(acc, pk[net * pn + (size_t)K * nf + j], pk[net * pn + (size_t)K * nf + nf + j]);
            hs[o] = y / (1.f + expf(-y));
        }
        lds_barrier();
        // ---- second map: (part of the hidden range, net, i): W1t [nf][K] against the activations of its net
        map_phase<RB, V>(pk + (size_t)K * nf + 2 * nf, 2 * K, nf, P, hs, 2 * nf, 1, 0, K, nf, pn, part, tid);
        lds_barrier();
        // ---- mu, logvar and the affine update of the warped coordinates: one (row, i) per thread
        const float *b1m = pk + (size_t)2 * K * nf + 2 * nf, *b1l = b1m + pn;
        for (int o = tid; o < RB * K; o += THREADS) {
            const int r = o / K, i = o - r * K, row = row0 + r;
            float om = o < THREADS ? b_mu : b1m[i], ol = o < THREADS ? b_lv : b1l[i];
            for (int p = 0; p < P; ++p) {
                om += part[(p * RB + r) * 2 * K + i];
                ol += part[(p * RB + r) * 2 * K + K + i];
            }
            const float lv = logf(a.eps + expf(ol));
            const int wi = kmul * i + wadd, ki = kmul * i + kadd;
            const float gold = gcur[r * G + wi];
            gcur[r * G + wi] = a.inverse ? expf(-0.5f * lv) * (gold - om) : fmaf(expf(0.5f * lv), gold, om);
            tot[r * G + wi] += lv;
            if (row < a.B) {
                const size_t base = ((size_t)s * a.B + row) * G;
                if (a.mus) { a.mus[base + wi] = om; a.mus[base + ki] = 0.f; }
                if (a.lvs) { a.lvs[base + wi] = lv; a.lvs[base + ki] = 0.f; }
            }
        }
        lds_barrier();
        if (a.gs)
            for (int e = tid; e < RB * G; e += THREADS) {
                const int r = e / G, row = row0 + r;
                if (row < a.B) a.gs[((size_t)s * a.B + row) * G + (e - r * G)] = gcur[e];
            }
    }
    for (int e = tid; e < RB * G; e += THREADS) {
        const int r = e / G, row = row0 + r;
        if (row >= a.B) continue;
        if (a.g_out) a.g_out[(size_t)row * G + (e - r * G)] = gcur[e];
        if (a.sum_lv) a.sum_lv[(size_t)row * G + (e - r * G)] = tot[e];
    }
}

}  // namespace

extern "C" {

size_t dpf_gprior_canon_floats(int G, int n_features) { return G > 0 && n_features > 0 ? 2 * canon_net_floats(G / 2, n_features) : 0; }

size_t dpf_gprior_packed_floats(int n_steps, int G, int n_features) {
    return n_steps > 0 && G > 0 && n_features > 0 ? (size_t)n_steps * 2 * packed_net_floats(G / 2, n_features) : 0;
}

int dpf_gprior_pack(int n_steps, int G, int n_features, float bn_eps, const float *canon, float *packed, dpf_stream_t stream) {
    if (n_steps <= 0 || G < 2 || (G & 1) || n_features <= 0 || !canon || !packed) return DPF_EINVAL;
    const size_t total = dpf_gprior_packed_floats(n_steps, G, n_features);
    const int blocks = (int)((total + PACK_THREADS - 1) / PACK_THREADS < 2048 ? (total + PACK_THREADS - 1) / PACK_THREADS : 2048);
    hipLaunchKernelGGL(gprior_pack_kernel, dim3(blocks), dim3(PACK_THREADS), 0, (hipStream_t)stream, 2 * n_steps, G / 2, n_features, bn_eps, canon,
                       packed);
    return (int)hipGetLastError();
}

int dpf_gprior_forward(int n_steps, int B, int G, int n_features, int mode, const int *codes, const float *packed, const float *g,
                       float *gs, float *mus, float *lvs, float *sum_lv, float *g_out, float eps, dpf_stream_t stream) {
    if (n_steps <= 0 || n_steps > MAX_STEPS || B < 0 || G < 2 || (G & 1) || n_features <= 0 || !codes || !packed || (mode != 0 && mode != 1))
        return DPF_EINVAL;
    if (B == 0) return 0;
    if (!g) return DPF_EINVAL;
    GArgs a = {};
    a.S = n_steps; a.B = B; a.G = G; a.nf = n_features; a.inverse = mode; a.eps = eps;
    a.packed = packed; a.g = g; a.gs = gs; a.mus = mus; a.lvs = lvs; a.sum_lv = sum_lv; a.g_out = g_out;
    for (int s = 0; s < n_steps; ++s) {
        if (codes[s] < 0 || codes[s] > 3) return DPF_EINVAL;
        a.codes[s >> 4] |= (uint32_t)codes[s] << ((s & 15) * 2);
    }
    // rows per workgroup: 1 up to one workgroup per CU, 2 beyond (the weights are read once per workgroup and step);
    // the hidden range of the second map is split over the threads the (net, i) items leave idle
    const int rb = B > 256 ? 2 : 1;
    const int K = G / 2;
    // V consecutive outputs per thread (16-byte weight loads) when the shapes allow it; the inner ranges are split so
    // that all 1024 threads have an item
    const int v = (K % 4 == 0 && n_features % 4 == 0) ? 4 : 1;
    const int q1 = 2 * n_features / v, q2 = G / v;
    a.parts1 = THREADS / q1 > 1 ? (THREADS / q1 < K ? THREADS / q1 : K) : 1;
    a.parts2 = THREADS / q2 > 1 ? (THREADS / q2 < n_features ? THREADS / q2 : n_features) : 1;
    const size_t p1 = (size_t)a.parts1 * 2 * n_features, p2 = (size_t)a.parts2 * G;
    const size_t lds = sizeof(float) * rb * ((size_t)2 * G + 2 * n_features + (p1 > p2 ? p1 : p2));
    if (lds > 64 * 1024) return DPF_ENOSUP;
    const int grid = (B + rb - 1) / rb;
    void (*kern)(GArgs) = rb == 1 ? (v == 4 ? gprior_kernel<1, 4> : gprior_kernel<1, 1>) : (v == 4 ? gprior_kernel<2, 4> : gprior_kernel<2, 1>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
