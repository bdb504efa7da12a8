// Fused per-point conditional affine-coupling flow stack for gfx950 (MI355X).
//
// Replaces, for eval-mode BatchNorm, the whole of
//   LocalCondRNVPDecoder.forward      lib/networks/decoders.py:54-72
//   CondRealNVPFlow3DTriple.forward   lib/networks/flows.py:151-160
//   CondRealNVPFlow3D.forward         lib/networks/flows.py:95-117
//   SharedDot.forward                 lib/networks/layers.py:40-45
// (~30 ATen kernels and ~7.7 KB of HBM traffic per point per layer in the
// reference) by THREE kernels:
//
//   pack_kernel   once per weight version: folds BatchNorm into the SharedDot
//                 weights, splits them into bf16 parts and lays them out in
//                 MFMA-fragment order;
//   film_kernel   once per batch: the per-cloud FiLM conditioner MLPs of every
//                 layer (flows.py:33-45,68-80), folded with BN1 and the output
//                 SharedDot into three per-cloud vectors per branch;
//   flow_kernel   the L-layer stack.  A wave owns 32 contiguous points of one
//                 cloud for ALL layers: the points, their running sum of
//                 log-variances and every activation stay in registers; HBM
//                 sees 12 B/point in, 24 B/point out (+36 B/point/layer only
//                 if the caller asks for the per-layer lists).
//
// Per layer and branch, for a 32-point tile (F = 64 hidden features):
//   h0 = relu(BN0(W0 x))      one  v_mfma_f32_32x32x16_bf16 per 32 features: the
//                             K=16 slots hold the 3-way bf16 split of the two
//                             inputs and of the folded weights/bias, so the
//                             result is fp32-accurate;
//   h1 = W1 h0                2 x 4 x {1,3,6} v_mfma_f32_32x32x16_bf16 (bf16,
//                             bf16x3 or bf16x6 split precision), fp32 accumulate.
//                             The accumulator starts at the folded FiLM shift.
//   o  = W2' relu(h1 + D)     VALU on the accumulator fragment + one
//                             cross-half swap (v_permlane32_swap).
// Weights are the A operand (rows = output features) and points the B operand
// (columns), so the C/D fragment of one MFMA -- lane holds 16 features of ITS
// point -- is, after relu and a bf16 split in registers, directly the B
// fragment of the next MFMA: the K-slot <-> feature permutation this implies
// is applied to W1's columns at pack time.  No LDS round trip for activations.
// LDS holds the current and the next layer's packed weights (double-buffered,
// streamed with global_load_lds while the current layer computes).
#include <stdlib.h>

#include "flow_common.h"

// -DDPF_ABLATE=<bitmask>: timing experiments of tools/ab_run.py (results are garbage): 1 no relu/split VALU, 2 no output
// contraction, 4 one chain MFMA per k-step, 8 no fragment reads, 16 no weight DMA, 32 no workgroup barriers, 64 no
// transcendental coupling transform, 128 no conditioner at all, 256 no input MFMAs (G0), 512 no accumulator init from LDS,
// 1024 half the A-fragment reads (the second M tile reuses the first's)
#ifndef DPF_ABLATE
#define DPF_ABLATE 0
#endif

namespace {

// ---- canonical fp32 layout (see dpf_hip.h) --------------------------------
constexpr int C_W0 = 0;
constexpr int C_BN0 = 128;     // gamma, beta, rm, rv
constexpr int C_W1 = 384;
constexpr int C_BN1 = 4480;    // rm, rv
constexpr int C_W2 = 4608;     // [2][64]
constexpr int C_B2 = 4736;     // [4]
constexpr int C_FILM = 4740;
__host__ __device__ constexpr int c_film_floats(int G) { return 64 * G + 256 + 4096 + 64; }
__host__ __device__ constexpr int c_branch_floats(int G) { return C_FILM + 2 * c_film_floats(G); }
__host__ __device__ constexpr int c_layer_floats(int G) { return 2 * c_branch_floats(G); }

// ===========================================================================
// pack
// ===========================================================================
template <int NS, bool F16 = false>
__global__ __launch_bounds__(256) void pack_kernel(int G, const float *__restrict__ canon, uint8_t *__restrict__ packed) {
    const int l = blockIdx.x;
    const float *cl = canon + (size_t)l * c_layer_floats(G);
    uint8_t *out = packed + (size_t)l * p_layer_bytes(NS);
    uint16_t *o16 = (uint16_t *)out;
    __shared__ float scratch[256];
    float wsc[2] = {1.0f, 1.0f};                  // F16: the power of two of each branch's W1 (flow_common.h: w1_pow2_scale)
    if (F16) {
        wsc[0] = w1_pow2_scale(cl + 0 * c_branch_floats(G) + C_W1, scratch);
        wsc[1] = w1_pow2_scale(cl + 1 * c_branch_floats(G) + C_W1, scratch);
    }
    // A1: W1 in fragment order.  Element j of lane (i, h), k-step s, M-tile t':
    //   W1[32t'+i][feat(s, j, h)],  feat = 32*(s>>1) + (r&3) + 8*(r>>2) + 4h,  r = 8*(s&1) + j
    // -- the feature a lane of the h0 accumulator fragment holds in register r of tile s>>1.
    for (int idx = threadIdx.x; idx < 2 * 2 * 4 * 64 * 8; idx += blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 3, tp = (idx >> 11) & 1, br = idx >> 12;
        const int i = lane & 31, h = lane >> 5;
        const int r = 8 * (s & 1) + j;
        const int fi = 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float w = cl[br * c_branch_floats(G) + C_W1 + (32 * tp + i) * 64 + fi] * (br ? wsc[1] : wsc[0]);
        if (F16) {                     // fp16 hi (RNE) + fp16 of the exact remainder: 22 significant bits
            const _Float16 wh = (_Float16)w;
            const _Float16 wl = (_Float16)(w - (float)wh);
            o16[idx] = __builtin_bit_cast(uint16_t, wh);
            o16[P_A1_PART / 2 + idx] = __builtin_bit_cast(uint16_t, wl);
        } else if (NS == 1) {
            o16[idx] = (uint16_t)bf16_rne(w);
        } else if (NS == 2) {
            float r1;
            o16[idx] = (uint16_t)(split_hi(w, r1) >> 16);
            o16[P_A1_PART / 2 + idx] = (uint16_t)bf16_rne(r1);
        } else {
            float r1, r2;
            o16[idx] = (uint16_t)(split_hi(w, r1) >> 16);
            o16[P_A1_PART / 2 + idx] = (uint16_t)(split_hi(r1, r2) >> 16);
            o16[P_A1_PART + idx] = (uint16_t)bf16_rne(r2);
        }
    }
    // A0: BN0-folded first SharedDot + bias, 3-way split over the 16 K slots of one MFMA.
    //   half h (0: keep channel a, 1: keep channel b) slots j:
    //   A = [wh wh wm wh wm wl | T*]   B = [xh xm xh xl xm xh | 1 (1|0)]
    //   T* = (Th, Tm) for h = 0 and (Tl, 0) for h = 1.
    uint16_t *a0 = (uint16_t *)(out + p_a0_off(NS));
    for (int idx = threadIdx.x; idx < 2 * 2 * 64 * 8; idx += blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) & 1, br = idx >> 10;
        const int f = 32 * t + (lane & 31), h = lane >> 5;
        const float *cb = cl + br * c_branch_floats(G);
        const float gamma = cb[C_BN0 + f], beta = cb[C_BN0 + 64 + f], rm = cb[C_BN0 + 128 + f], rv = cb[C_BN0 + 192 + f];
        const float s0 = gamma / sqrtf(rv + BN_EPS);
        const float w = s0 * cb[C_W0 + f * 2 + h];
        const float T = beta - rm * s0;
        const uint32_t v = input_weight_slot(w, T, h, j);
        a0[idx] = (uint16_t)v;
    }
}

// ===========================================================================
// FiLM conditioner
// ===========================================================================
// fp32 conditioner weights, transposed at pack time so that the 64 output features are the
// contiguous (lane) dimension: per (layer, branch, sub-net w|b):
//   WT[G][64] | sc[64] sh[64] (eval BatchNorm folded: u*sc + sh) | W1T[64][64] | bf1[64]
// and per (layer, branch): s1[64] t1[64] (BN1, affine=False) w2a[64] w2b[64] (output SharedDot rows).
__host__ __device__ constexpr int fw_sub_floats(int G) { return 64 * G + 128 + 4096 + 64; }
__host__ __device__ constexpr size_t fw_total_floats(int L, int G) { return (size_t)L * 4 * fw_sub_floats(G) + (size_t)L * 2 * 320; }

__global__ __launch_bounds__(256) void pack_film_kernel(int L, int G, const float *__restrict__ canon, float *__restrict__ fw, int f16) {
    const int l = blockIdx.x >> 2, br = (blockIdx.x >> 1) & 1, sub = blockIdx.x & 1;
    const float *cb = canon + (size_t)l * c_layer_floats(G) + br * c_branch_floats(G);
    __shared__ float scratch[256];
    // f16x3: the branch's W1 is packed times 2^k (pack_kernel) -- the output SharedDot's rows carry 2^-k from here on, D gets 2^k in film_kernel
    const float wsc = (f16 && sub == 0) ? w1_pow2_scale(cb + C_W1, scratch) : 1.0f;
    const float *cf = cb + C_FILM + sub * c_film_floats(G);
    const float *Wf0 = cf, *bnf = cf + 64 * G, *Wf1 = bnf + 256, *bf1 = Wf1 + 4096;
    float *o = fw + (size_t)blockIdx.x * fw_sub_floats(G);
    for (int idx = threadIdx.x; idx < 64 * G; idx += 256) o[idx] = Wf0[(idx & 63) * G + (idx >> 6)];
    o += 64 * G;
    if (threadIdx.x < 64) {
        const int f = threadIdx.x;
        const float sc = bnf[f] / sqrtf(bnf[192 + f] + BN_EPS);
        o[f] = sc;
        o[64 + f] = bnf[64 + f] - bnf[128 + f] * sc;
        o[128 + 4096 + f] = bf1[f];
    }
    for (int idx = threadIdx.x; idx < 4096; idx += 256) o[128 + idx] = Wf1[(idx & 63) * 64 + (idx >> 6)];
    if (sub == 0 && threadIdx.x < 64) {
        const int f = threadIdx.x;
        float *c = fw + (size_t)L * 4 * fw_sub_floats(G) + (size_t)(l * 2 + br) * 320;
        const float s1 = 1.0f / sqrtf(cb[C_BN1 + 64 + f] + BN_EPS);
        c[f] = s1;
        c[64 + f] = -cb[C_BN1 + f] * s1;
        c[128 + f] = cb[C_W2 + f] * (1.0f / wsc);
        c[192 + f] = cb[C_W2 + 64 + f] * (1.0f / wsc);
        if (f < 2) c[256 + f] = cb[C_B2 + f];
        if (f == 2) c[258] = wsc;
    }
}

constexpr int FILM_CLOUDS = 8;    // clouds per workgroup

// 512 threads = 2 sub-nets (w, b) x 64 output features x 4 K-quarters.  Each thread reduces ONE
// quarter of the contraction for all 8 clouds, so its weight loads (coalesced: feature = lane)
// are all issued up front -- the kernel is two global round trips and two LDS reductions deep.
__global__ __launch_bounds__(512) void film_kernel(int L, int B, int G, const float *__restrict__ fw,
                                                   const float *__restrict__ g, float *__restrict__ film, float flow_eps) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float *gs = fsm;                                  // [8][G]
    float *part = fsm + FILM_CLOUDS * G;              // [2 sub][4 kq][8 clouds][64]
    float *hid = part + 2 * 4 * FILM_CLOUDS * 64;     // [2][8][64]
    float *cbx = hid + 2 * FILM_CLOUDS * 64;          // [8][64]
    const int l = blockIdx.x >> 1, br = blockIdx.x & 1;
    const int b0 = blockIdx.y * FILM_CLOUDS;
    const int tid = threadIdx.x, sub = tid >> 8, f = tid & 63, kq = (tid >> 6) & 3;
    const float *w = fw + (size_t)((l * 2 + br) * 2 + sub) * fw_sub_floats(G);
    const float *WT = w, *sc = w + 64 * G, *sh = sc + 64, *W1T = sh + 64, *bf1 = W1T + 4096;
    for (int e = tid; e < FILM_CLOUDS * G / 4; e += 512) {            // g rows of this workgroup's clouds
        const int row = e / (G / 4), c4 = (e % (G / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b0 + row < B) v = *(const f32x4 *)(g + (size_t)(b0 + row) * G + c4);
        *(f32x4 *)(gs + row * G + c4) = v;
    }
    // every small per-feature constant is requested now: nothing below waits on a fresh global round trip
    const float *cst = fw + (size_t)L * 4 * fw_sub_floats(G) + (size_t)(l * 2 + br) * 320;
    const float bn_a = sc[f], bn_d = sh[f], bias1 = bf1[f];
    const float s1 = cst[f], t1 = cst[64 + f], w2a = cst[128 + f], w2b = cst[192 + f], b2v = cst[256 + (f & 1)];
    const float wsc = cst[258];                  // f16x3: 2^k of this branch's packed W1 (1 otherwise); w2a / w2b already carry 2^-k
    float acc[FILM_CLOUDS];
#pragma unroll
    for (int c = 0; c < FILM_CLOUDS; ++c) acc[c] = 0.f;
    const int kspan = G / 4, k0 = kq * kspan;                         // G % 16 == 0
    for (int kk = 0; kk < kspan; kk += 32) {                          // u = g . Wf0^T        flows.py:34/41
        float wv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) wv[i] = kk + i < kspan ? WT[(size_t)(k0 + kk + i) * 64 + f] : 0.f;
        if (kk == 0) __syncthreads();                                 // gs visible (loads above already in flight)
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            if (kk + i < kspan) {
#pragma unroll
                for (int c = 0; c < FILM_CLOUDS; ++c) {
                    const f32x4 x = *(const f32x4 *)(gs + c * G + k0 + kk + i);
                    // explicit fma chain: the same bits for a cloud wherever it sits in the batch
                    acc[c] = __builtin_fmaf(wv[i + 3], x.w, __builtin_fmaf(wv[i + 2], x.z, __builtin_fmaf(wv[i + 1], x.y, __builtin_fmaf(wv[i], x.x, acc[c]))));
                }
            }
        }
    }
    float w1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w1[i] = W1T[(kq * 16 + i) * 64 + f];  // second layer's weights: in flight early
#pragma unroll
    for (int c = 0; c < FILM_CLOUDS; ++c) part[((sub * 4 + kq) * FILM_CLOUDS + c) * 64 + f] = acc[c];
    __syncthreads();
    {   // BatchNorm1d over the batch dim in eval mode (flows.py:35/42), then Swish (layers.py:9-10)
        const float a = bn_a, d = bn_d;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = kq * 2 + cc;
            float u = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) u += part[((sub * 4 + q) * FILM_CLOUDS + c) * 64 + f];
            u = u * a + d;
            hid[(sub * FILM_CLOUDS + c) * 64 + f] = u / (1.0f + expf(-u));
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < FILM_CLOUDS; ++c) {                           // Linear(F, F)          flows.py:37/44
        const float *h0 = hid + (sub * FILM_CLOUDS + c) * 64 + kq * 16;
        float a2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const f32x4 x = *(const f32x4 *)(h0 + i);
            a2 = __builtin_fmaf(w1[i + 3], x.w, __builtin_fmaf(w1[i + 2], x.z, __builtin_fmaf(w1[i + 1], x.y, __builtin_fmaf(w1[i], x.x, a2))));
        }
        acc[c] = a2;
    }
    __syncthreads();                                                  // everyone is done reading part (phase 1)
#pragma unroll
    for (int c = 0; c < FILM_CLOUDS; ++c) part[((sub * 4 + kq) * FILM_CLOUDS + c) * 64 + f] = acc[c];
    __syncthreads();
    float v[2];
    {
        const float bias = bias1;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = kq * 2 + cc;
            float t = bias;
#pragma unroll
            for (int q = 0; q < 4; ++q) t += part[((sub * 4 + q) * FILM_CLOUDS + c) * 64 + f];
            v[cc] = t;
            if (sub == 1) cbx[c * 64 + f] = t;
        }
    }
    __syncthreads();
    if (sub == 1) return;
    // fold FiLM (flows.py:100-101) with BN1 (affine=False, :30/65) and the output SharedDot (:49/84):
    //   relu((eps+e^cw) * BN1(h1) + cb) = FA * relu(h1 + FC/FA),  FA = (eps+e^cw)/sqrt(rv1+eps_bn) > 0
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = kq * 2 + cc, b = b0 + c;
        if (b >= B) continue;
        const float a = flow_eps + expf(v[cc]);
        const float FA = a * s1, FC = a * t1 + cbx[c * 64 + f];
        float *o = film + ((size_t)l * B + b) * (FILM_BYTES / 4) + br * FILM_BR_FLOATS;
        o[f] = (FC / FA) * wsc;
        o[64 + f] = w2a * FA;
        o[128 + f] = w2b * FA;
        if (f < 2) film[((size_t)l * B + b) * (FILM_BYTES / 4) + FILM_B2_OFF + br * 2 + f] = b2v;
    }
}

// ===========================================================================
// the fused stack
// ===========================================================================
// Stream one layer (packed weights + this cloud's FiLM vectors) into an LDS
// buffer: 1 KiB per wave-instruction, straight to LDS (no VGPR staging).
// Issuing a piece blocks the issuing wave for ~60-180 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost").  In an
// 8-wave workgroup the SIMD's arbiter favours the older wave of each pair, so waves 0-3 run ahead and idle at the
// workgroup barrier (r02 phase profile: ~4300 cycles per two layers) while waves 4-7 are the critical path: the leaders
// issue ALL pieces (LW = FW / 2 issuing waves) and the laggards none.
template <int K, int PER>
__device__ __forceinline__ void issue_pieces(const uint8_t *src, uint8_t *dst) {
    if constexpr (K < PER) {
        __builtin_amdgcn_global_load_lds((glb_void *)(src + (K / 4) * 4096), (lds_void *)(dst + (K / 4) * 4096), 16, (K % 4) * 1024, 0);
        issue_pieces<K + 1, PER>(src, dst);
    }
}
template <int NS, int FW>
__device__ __forceinline__ void stage_layer(const FlowArgs &a, int li, int bi, uint8_t *lds, int wave, int lane) {
    constexpr int NP = p_layer_bytes(NS) / 1024, NF = FILM_BYTES / 1024;
    constexpr int LW = FW >= 8 ? FW / 2 : FW, PER = NP / LW;
    static_assert(NP % LW == 0, "every issuing wave takes the same contiguous run of weight pieces");
    if ((unsigned)wave >= (unsigned)LW) return;          // `wave` = index among the issuing waves (negative: not an issuer)
    // a wave's pieces are contiguous in memory and in LDS, so one (address, M0) pair serves four pieces through the
    // instruction's immediate offset (it advances both sides): ~1.5 instructions per piece instead of ~18
    const uint8_t *src = a.packed + (size_t)li * p_layer_bytes(NS) + wave * (PER * 1024) + lane * 16;
    uint8_t *dst = lds + wave * (PER * 1024);
    issue_pieces<0, PER>(src, dst);
#pragma unroll
    for (int f = 0; f < NF; ++f)
        if (wave == f % LW) {
            const uint8_t *fsrc = (const uint8_t *)a.film + ((size_t)li * a.B + bi) * FILM_BYTES + f * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((glb_void *)fsrc, (lds_void *)(lds + (NP + f) * 1024), 16, 0, 0);
        }
}

#ifdef DPF_PROFILE
#define DPF_T(i) { __builtin_amdgcn_sched_barrier(0); tt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define DPF_T(i)
#endif
template <int NS, bool TWO>
__device__ __forceinline__ void branch_tile(const uint8_t *lb, int br, int lane, int h, u32x4 b0, float &oa, float &ob, unsigned long long *tt) {
    constexpr int A0OFF = p_a0_off(NS), FILMOFF = p_layer_bytes(NS);
    typedef Terms<NS> TT;
    // ---- h0 = relu(BN0(W0 x)) on the matrix core, fp32-accurate
    u32x4 bfrag[NS][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const u32x4 a0 = *(const u32x4 *)(lb + A0OFF + ((br * 2 + t) * 64 + lane) * 16);
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc0 = mfma(a0, b0, acc0);
        // relu + bf16 split; accumulator register r of tile t is element j = r&7 of k-step 2t + (r>>3)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = relu(acc0[r]), v1 = relu(acc0[r + 1]);
            const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;
            if (NS == 1) {
                bfrag[0][s][d] = pack_bf16_rne(v0, v1);
            } else if (NS == 2) {
                float l0, l1;
                split_hi(v0, l0); split_hi(v1, l1);
                bfrag[0][s][d] = pack_bf16_trunc(v0, v1);
                bfrag[1][s][d] = pack_bf16_rne(l0, l1);
            } else {
                float l0, l1, m0, m1;
                split_hi(v0, l0); split_hi(v1, l1);
                split_hi(l0, m0); split_hi(l1, m1);
                bfrag[0][s][d] = pack_bf16_trunc(v0, v1);
                bfrag[1][s][d] = pack_bf16_trunc(l0, l1);
                bfrag[2][s][d] = pack_bf16_rne(m0, m1);
            }
        }
    }
    DPF_T(1)
    // ---- h1 = W1 h0, accumulator pre-loaded with the folded FiLM shift D.
    // The A fragments stream from LDS in batches of 4 (two k-steps x two M tiles), one batch
    // ahead of the MFMAs that consume them (hipcc otherwise reuses ONE fragment register and
    // waits for each ds_read before each MFMA).
    const float *fl = (const float *)(lb + FILMOFF) + br * FILM_BR_FLOATS;
    f32x16 acc1[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 dv = *(const f32x4 *)(fl + 32 * tp + 8 * q + 4 * h);
            acc1[tp][4 * q + 0] = dv.x; acc1[tp][4 * q + 1] = dv.y;
            acc1[tp][4 * q + 2] = dv.z; acc1[tp][4 * q + 3] = dv.w;
        }
    // One batch = the NS parts of both M tiles of ONE k-step (2*NS fragments); every fragment feeds all the
    // product terms that use its part (bf16x3: the hi part twice) -- an LDS->VGPR fragment load costs the SIMD
    // ~18 cycles next to the 32 of an MFMA (tools/ubench/lds_mfma.hip), so loads are not repeated per term.
    // Batches run one ahead of the MFMAs that consume them.
    u32x4 af[2][2 * NS];
    auto load_batch = [&](int ks, u32x4 (&dst)[2 * NS]) {
#pragma unroll
        for (int part = 0; part < NS; ++part)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                if ((DPF_ABLATE & 1024) && tp == 1) { dst[part * 2 + 1] = dst[part * 2]; continue; }   // timing: half the fragment reads
                dst[part * 2 + tp] = *(const u32x4 *)(lb + part * P_A1_PART + (((br * 2 + tp) * 4 + ks) * 64 + lane) * 16);
            }
    };
    load_batch(0, af[0]);
    const float *wa = fl + 64, *wb2 = fl + 128;
    f32x4 wva[2][4], wvb[2][4];                    // epilogue weights, one M tile at a time
    auto load_w = [&](int tp, f32x4 (&da)[4], f32x4 (&db)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * tp + 8 * q + 4 * h;
            da[q] = *(const f32x4 *)(wa + f0);
            if (TWO) db[q] = *(const f32x4 *)(wb2 + f0);
        }
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) load_batch(ks + 1, af[(ks + 1) & 1]);
        else load_w(0, wva[0], wvb[0]);            // rides under the last batch of MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                acc1[tp] = mfma(af[ks & 1][TT::A[term] * 2 + tp], bfrag[TT::B[term]][ks], acc1[tp]);
        __builtin_amdgcn_sched_barrier(0);
    }
    DPF_T(2)
    // ---- o = W2' relu(h1 + D): each lane reduces its 32 features.  Written as packed-f32 FMAs on (even, odd)
    // partial sums over NATURAL register pairs -- two consecutive accumulator registers against the two consecutive
    // weights of one f32x4 load -- so that no operand pair has to be assembled with v_mov (the SLP vectoriser paired
    // the scalar chain across the two M tiles and spent ~1.5 v_mov per v_pk_fma_f32 doing so).
    // (r03: scalar again, as in layer_pipe -- a packed-f32 instruction costs +16 cycles beside the SIMD partner's MFMAs;
    // two partial sums per output keep the rounding order of the packed form: even / odd registers)
    float oa0 = 0.f, oa1 = 0.f, ob0 = 0.f, ob1 = 0.f;
    load_w(1, wva[1], wvb[1]);
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = relu(acc1[tp][4 * q + 0]), v1 = relu(acc1[tp][4 * q + 1]);
            const float v2 = relu(acc1[tp][4 * q + 2]), v3 = relu(acc1[tp][4 * q + 3]);
            const f32x4 wa4 = wva[tp][q];
            oa0 = __builtin_fmaf(wa4.x, v0, oa0); oa1 = __builtin_fmaf(wa4.y, v1, oa1);
            oa0 = __builtin_fmaf(wa4.z, v2, oa0); oa1 = __builtin_fmaf(wa4.w, v3, oa1);
            if (TWO) {
                const f32x4 wb4 = wvb[tp][q];
                ob0 = __builtin_fmaf(wb4.x, v0, ob0); ob1 = __builtin_fmaf(wb4.y, v1, ob1);
                ob0 = __builtin_fmaf(wb4.z, v2, ob0); ob1 = __builtin_fmaf(wb4.w, v3, ob1);
            }
        }
    oa = oa0 + oa1;
    ob = ob0 + ob1;
}

// ---------------------------------------------------------------------------------------------------------
// Both conditioner branches of one layer for one 32-point tile as ONE software pipeline (NS <= 2).
//
// Measured on gfx950 (tools/ubench/mfma_fill.hip, profiles/r02_mfma_fill.txt): a v_mfma_f32_32x32x16_bf16 occupies the
// SIMD's matrix pipe for 32 cycles, and up to SIX plain VALU instructions issued behind it are free (32.3 cycles per
// MFMA with 6 fillers, whether the fillers come from the same wave or from the other wave of the SIMD); every further
// one costs ~4.3 cycles, and a packed-f32 instruction (v_pk_fma/add/mul_f32) costs +16 cycles -- so this file is
// compiled with -fno-slp-vectorize and the contraction below is scalar.  The two branches (logvar, mu) depend on the
// layer input only, so their stages interleave: the VALU work of one k-step (relu + bf16 hi/lo split of 8 accumulator
// registers: 32 instructions) and the LDS fragment reads of the next k-step ride in the issue gaps of the 6 MFMAs of
// the current k-step.  Every `group` below is one scheduling region (fenced by sched_barrier) holding <= 6 MFMAs and
// the VALU / DS work to hide behind them; sched_group_barrier pins the MFMA : VALU : DS-read order inside it.
//
//   G0      4 input MFMAs (A t0, A t1, B t0, B t1)          | split A k0
//   G1..G3  chain A k0..k2 (6 MFMAs each at bf16x3)          | split A k1..k3, A fragments of the next k-step, D of B
//   G4      chain A k3                                       | split B k0
//   G5..G7  chain B k0..k2                                   | split B k1..k3, B fragments, output weights of A
//   G8      chain B k3                                       | output contraction of A
//   tail    output contraction of B
template <int NS, bool TWO, bool F16>
__device__ __forceinline__ void layer_pipe(const uint8_t *lb, int lane, int h, u32x4 b0, float negone, int midbar,
                                           float (&o)[2][2], unsigned long long *tt) {
    static_assert(!F16 || NS == 2, "fp16 operands: hi/lo split only");
    static_assert(NS <= 2, "the pipelined body keeps both branches' fragments in registers: bf16 / bf16x3 only");
    constexpr int A0OFF = p_a0_off(NS), FILMOFF = p_layer_bytes(NS);
    typedef Terms<NS> TT;
    const float *film = (const float *)(lb + FILMOFF);
    // The FiLM vectors' reads are "slot + constant + 16 h": left alone the compiler keeps the lane part (16 h) in one register,
    // puts slot + constant into 32 SGPRs and spends a v_add_u32 per read -- 32 VALU per layer in a kernel bound by VALU issue.
    // One LDS pointer per layer, opaque to the optimiser, makes every read that register + a 16-bit immediate (r04).
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    typedef __attribute__((address_space(3))) const f32x4 lds_f4;
    lds_u8 *fl3 = (lds_u8 *)(lb + FILMOFF + 16 * h);
    asm volatile("" : "+v"(fl3));
    auto film4 = [&](int float_index) { return *(lds_f4 *)(fl3 + 4 * float_index); };      // floats film[i + 4 h .. i + 4 h + 3]
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc0[2][2], acc1[2][2];
    u32x4 bfrag[2][NS][4];
    u32x4 af[2][2 * NS];

    auto ld_frag = [&](int br, int ks, u32x4 (&dst)[2 * NS]) {     // the NS parts of both M tiles of k-step ks
        if ((DPF_ABLATE & 8) && !(br == 0 && ks == 0)) {   // timing experiment: no fragment reads after the first
#pragma unroll
            for (int i = 0; i < 2 * NS; ++i) dst[i] = af[0][i] + (uint32_t)ks;
            return;
        }
#pragma unroll
        for (int part = 0; part < NS; ++part)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                dst[part * 2 + tp] = *(const u32x4 *)(lb + part * P_A1_PART + (((br * 2 + tp) * 4 + ks) * 64 + lane) * 16);
    };
    auto init_acc1 = [&](int br, int tp) {                          // accumulator starts at the folded FiLM shift D
        if (DPF_ABLATE & 512) { acc1[br][tp] = z16; return; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 dv = film4(br * FILM_BR_FLOATS + 32 * tp + 8 * q);
            acc1[br][tp][4 * q + 0] = dv.x; acc1[br][tp][4 * q + 1] = dv.y;
            acc1[br][tp][4 * q + 2] = dv.z; acc1[br][tp][4 * q + 3] = dv.w;
        }
    };
    // relu + bf16 split of the 8 accumulator registers that form the B fragment(s) of k-step ks: 32 VALU at bf16x3
    auto split_ks = [&](int br, int ks) {
        const int t = ks >> 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int r = 8 * (ks & 1) + 2 * d;
            if (DPF_ABLATE & 1) {          // timing experiment: no relu / split VALU
                bfrag[br][0][ks][d] = f2u(acc0[br][t][r]);
                if (NS > 1) bfrag[br][NS - 1][ks][d] = f2u(acc0[br][t][r + 1]);
                continue;
            }
            if (F16) {                     // fp16 hi/lo with the ReLU folded in: flow_common.h
                uint32_t hi_, lo_;
                split_relu_f16(acc0[br][t][r], acc0[br][t][r + 1], negone, hi_, lo_);
                bfrag[br][0][ks][d] = hi_;
                bfrag[br][1][ks][d] = lo_;
                continue;
            }
            const float v0 = relu(acc0[br][t][r]), v1 = relu(acc0[br][t][r + 1]);
            if (NS == 1) {
                bfrag[br][0][ks][d] = pack_bf16_rne(v0, v1);
            } else {
                float l0, l1;
                split_hi(v0, l0); split_hi(v1, l1);
                bfrag[br][0][ks][d] = pack_bf16_trunc(v0, v1);
                bfrag[br][NS - 1][ks][d] = pack_bf16_rne(l0, l1);
            }
        }
    };
    auto chain_ks = [&](int br, int ks, const u32x4 (&a)[2 * NS]) {
        if (DPF_ABLATE & 4) {              // timing experiment: one MFMA per k-step instead of 6
            acc1[br][0] = mfma(a[0], bfrag[br][0][ks], acc1[br][0]);
            acc1[br][1][0] += u2f(a[1].x ^ bfrag[br][NS - 1][ks].x);
            return;
        }
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                acc1[br][tp] = F16 ? mfma_f16(a[TT::A[term] * 2 + tp], bfrag[br][TT::B[term]][ks], acc1[br][tp])
                                   : mfma(a[TT::A[term] * 2 + tp], bfrag[br][TT::B[term]][ks], acc1[br][tp]);
    };
    // o = W2' relu(h1 + D) over this lane's 32 features: scalar FMAs on two partial sums per output (no packed f32).
    // The weights of one M tile are fetched one group ahead of the FMAs that use them.
    float pa[2][2], pb[2][2];                                      // [br][even/odd]
    f32x4 cwa[2][4], cwb[2][4];                                    // [slot][q]: slot = (br + tp) & 1 alternates
    auto ld_cw = [&](int br, int tp) {
        const int sl = (br * 2 + tp) & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = br * FILM_BR_FLOATS + 64 + 32 * tp + 8 * q;
            cwa[sl][q] = film4(f0);
            if (TWO) cwb[sl][q] = film4(f0 + 64);
        }
    };
    auto contract = [&](int br, int tp, int q0 = 0, int q1 = 4) {   // features 8 q0 .. 8 q1 of M tile tp
        const int sl = (br * 2 + tp) & 1;
        if (DPF_ABLATE & 2) {              // timing experiment: no output contraction VALU
            if (q0 == 0) { pa[br][0] += acc1[br][tp][0] + cwa[sl][0].x; pb[br][0] += acc1[br][tp][1]; }
            return;
        }
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const f32x4 wa4 = cwa[sl][q];
            const float r0 = relu(acc1[br][tp][4 * q + 0]), r1 = relu(acc1[br][tp][4 * q + 1]);
            const float r2 = relu(acc1[br][tp][4 * q + 2]), r3 = relu(acc1[br][tp][4 * q + 3]);
            pa[br][0] = __builtin_fmaf(wa4.x, r0, pa[br][0]); pa[br][1] = __builtin_fmaf(wa4.y, r1, pa[br][1]);
            pa[br][0] = __builtin_fmaf(wa4.z, r2, pa[br][0]); pa[br][1] = __builtin_fmaf(wa4.w, r3, pa[br][1]);
            if (TWO) {
                const f32x4 wb4 = cwb[sl][q];
                pb[br][0] = __builtin_fmaf(wb4.x, r0, pb[br][0]); pb[br][1] = __builtin_fmaf(wb4.y, r1, pb[br][1]);
                pb[br][0] = __builtin_fmaf(wb4.z, r2, pb[br][0]); pb[br][1] = __builtin_fmaf(wb4.w, r3, pb[br][1]);
            }
        }
    };
    {   // the output SharedDot's bias starts the sums: half of it in each lane half (exact halving; the halves are added later)
        const float *b2 = film + FILM_B2_OFF;
        for (int br = 0; br < 2; ++br) {
            pa[br][0] = 0.5f * b2[br * 2]; pa[br][1] = 0.f;
            pb[br][0] = TWO ? 0.5f * b2[br * 2 + 1] : 0.f; pb[br][1] = 0.f;
        }
    }

    // sched_barrier fences the machine scheduler only: pure arithmetic (VALU, MFMA) still drifts across it when the
    // selection DAG is linearised.  An empty volatile asm that "rewrites" a value is ordered with the fences (both have
    // side effects), so pinning a group's VALU INPUTS at its top and its OUTPUTS at its bottom keeps the work inside.
    auto pin = [](auto &x) { asm volatile("" : "+v"(x)); };
    auto pin_acc0 = [&](int br, int ks) {                          // the 8 registers split_ks(br, ks) reads
        const int t = ks >> 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) { float v = acc0[br][t][8 * (ks & 1) + i]; pin(v); acc0[br][t][8 * (ks & 1) + i] = v; }
    };
    auto pin_bfrag = [&](int br, int ks) {
#pragma unroll
        for (int part = 0; part < NS; ++part) pin(bfrag[br][part][ks]);
    };
    auto pin_acc1 = [&](int br, int tp) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { float v = acc1[br][tp][i]; pin(v); acc1[br][tp][i] = v; }
    };
    auto pin_sums = [&](int br) { pin(pa[br][0]); pin(pa[br][1]); if (TWO) { pin(pb[br][0]); pin(pb[br][1]); } };
    // `n` MFMAs, each followed by (the first `ds` of them) `dsper` LDS reads and by `v` VALU: the reads a later group waits
    // for are issued at the top of this one
#define DPF_PIPE_PATTERN(n, v, ds, dsper)                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        if (i_ < (ds)) __builtin_amdgcn_sched_group_barrier(0x100, (dsper), 0);     \
        __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);                        \
    }
    constexpr int NM = 2 * TT::N;      // MFMAs per k-step

    // ---- G0: input MFMAs (h0 pre-activation, fp32-accurate), A's first split
    {
        const u32x4 a00 = *(const u32x4 *)(lb + A0OFF + ((0 * 2 + 0) * 64 + lane) * 16);
        const u32x4 a01 = *(const u32x4 *)(lb + A0OFF + ((0 * 2 + 1) * 64 + lane) * 16);
        const u32x4 a10 = *(const u32x4 *)(lb + A0OFF + ((1 * 2 + 0) * 64 + lane) * 16);
        const u32x4 a11 = *(const u32x4 *)(lb + A0OFF + ((1 * 2 + 1) * 64 + lane) * 16);
        ld_frag(0, 0, af[0]);
        if (DPF_ABLATE & 256) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[0][0][i] = u2f(b0.x + i); acc0[0][1][i] = u2f(b0.y + i); acc0[1][0][i] = u2f(b0.z + i); acc0[1][1][i] = u2f(b0.w ^ i); }
        } else {
        acc0[0][0] = mfma(a00, b0, z16);
        acc0[0][1] = mfma(a01, b0, z16);
        acc0[1][0] = mfma(a10, b0, z16);
        acc0[1][1] = mfma(a11, b0, z16);
        }
        split_ks(0, 0);
        pin_bfrag(0, 0);
        init_acc1(0, 0); init_acc1(0, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // input fragments, A's first k-step
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // D of A: what G1's first MFMAs wait for
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    DPF_T(1)
    // ---- G1..G4: chain A
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int sbr = ks + 1 < 4 ? 0 : 1, sks = (ks + 1) & 3;      // the split that rides in this group
        pin_acc0(sbr, sks);
        ld_frag(sbr, sks, af[(ks + 1) & 1]);
        if (ks == 1) init_acc1(1, 0);
        if (ks == 2) init_acc1(1, 1);
        if (ks == 3) ld_cw(0, 0);                                     // A's output weights, one group ahead of their use
        chain_ks(0, ks, af[ks & 1]);
        split_ks(sbr, sks);
        pin_bfrag(sbr, sks);
        DPF_PIPE_PATTERN(NM, 6, 4, 2)
        __builtin_amdgcn_sched_barrier(0);
        if (ks < 3 && midbar == ks + 1) __syncthreads();
#ifdef DPF_PROFILE
        if (ks == 0) DPF_T(8) else if (ks == 1) DPF_T(9) else if (ks == 2) DPF_T(10)
#endif
    }
    DPF_T(2)
    // skewed ring (flow_kernel<.., SKEW>): the lagging half of the workgroup meets the leading half's end-of-layer barrier
    // at the group boundary `midbar` (r02 sweep over all eight boundaries: 46.2 us at 5, 46.5-48.4 elsewhere), about half a
    // layer behind, so that one wave's MFMA chains run beside its SIMD partner's VALU-only stretch
    if (!(DPF_ABLATE & 32) && midbar == 4) __syncthreads();
    // ---- G5..G8: chain B; A's output contraction rides in the later groups
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) { pin_acc0(1, ks + 1); ld_frag(1, ks + 1, af[(ks + 1) & 1]); }
        // A's output contraction in four quarters, one per group of chain B (a split of B + a quarter = 36..44 VALU behind
        // 6 MFMAs at f16x3: at the free-filler budget; bf16x3's 32-VALU split runs ~2 over per gap)
        if (ks == 0) pin_acc1(0, 0);
        if (ks == 1) ld_cw(0, 1);
        if (ks == 2) pin_acc1(0, 1);
        if (ks == 3) ld_cw(1, 0);
        chain_ks(1, ks, af[ks & 1]);
        if (ks + 1 < 4) { split_ks(1, ks + 1); pin_bfrag(1, ks + 1); }
        contract(0, ks >> 1, 2 * (ks & 1), 2 * (ks & 1) + 2);
        pin_sums(0);
        if (ks < 3) { if (TWO) { DPF_PIPE_PATTERN(NM, (F16 ? 8 : 10), 4, 3) } else { DPF_PIPE_PATTERN(NM, (F16 ? 6 : 8), 4, 3) } }
        else { if (TWO) { DPF_PIPE_PATTERN(NM, 4, 4, 3) } else { DPF_PIPE_PATTERN(NM, 3, 4, 3) } }
        __builtin_amdgcn_sched_barrier(0);
        if (midbar == ks + 5) __syncthreads();
#ifdef DPF_PROFILE
        if (ks == 0) DPF_T(11) else if (ks == 1) DPF_T(12) else if (ks == 2) DPF_T(13)
#endif
    }
    DPF_T(3)
    // ---- tail: B's output contraction
    ld_cw(1, 1);
    contract(1, 0);
    contract(1, 1);
#undef DPF_PIPE_PATTERN
#pragma unroll
    for (int br = 0; br < 2; ++br) { o[br][0] = pa[br][0] + pa[br][1]; o[br][1] = pb[br][0] + pb[br][1]; }
}

__device__ __forceinline__ float half_sum(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    return u2f(r[0]) + u2f(r[1]);
}

// The fused L-layer stack.  FW waves per workgroup, each wave owns one 32-point tile of ONE
// cloud for all layers (both conditioner branches).  Every workgroup streams the layer weights
// through its own LDS, so bigger workgroups mean less L2->LDS traffic per point.
// LPB = layers per LDS buffer: with two layers per buffer (2 x 2 x 38 KiB at bf16x3: one workgroup per CU, which is
// all cfg-2 offers anyway) the workgroup barrier that hands a buffer over comes every other layer.
// SKEW (8-wave workgroups, pipelined body): three one-layer LDS buffers in a ring and a phase-locked half-layer skew
// between the two waves of every SIMD.  Waves 0-3 ("leaders") hit the workgroup barrier at the END of layer n and then
// issue the DMA of layer n + 2 into the buffer layer n - 1 used; waves 4-7 ("laggards") hit the same barrier in the MIDDLE
// of their layer n (between the two MFMA chains, layer_pipe's midbar).  Barrier n therefore completes when the leaders
// have finished layer n and the laggards half of it: every wave makes the same number of barrier calls, the skew is
// sustained by construction, and while one wave of a SIMD is in its MFMA chains (matrix pipe busy, <= 6 VALU per gap)
// its partner is in the VALU-only part of a layer (contraction tail, coupling transform, next input fragment, input
// MFMAs + first split).  Buffer safety: after barrier n nobody reads layer n - 1 any more (laggards are past the middle
// of layer n), and the DMA of layer n + 1, issued after barrier n - 1, was waited for (vmcnt 0) by its issuers before
// they arrived at barrier n.
template <int NS, int FW, int LPB = 1, bool PIPE = false, bool F16 = false, bool SKEW = false, bool INV = false>
__global__ __launch_bounds__(FW * 64, 2) void flow_kernel(FlowArgs a) {
    static_assert(!SKEW || (FW == 8 && LPB == 1 && PIPE), "the skewed ring is built for 8-wave pipelined workgroups");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int LBYTES = p_layer_bytes(NS) + FILM_BYTES;
    constexpr int FILMOFF = p_layer_bytes(NS);

    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, L = a.L;
    const int n = (blockIdx.x * FW + wave) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const size_t cloud = (size_t)bi * 3 * N;
    float p0 = a.p_in[cloud + nc], p1 = a.p_in[cloud + N + nc], p2 = a.p_in[cloud + 2 * (size_t)N + nc];
    if (a.base_mu != nullptr) {            // reparameterize: eps.mul(exp(0.5 * logvar)).add_(mu), every op rounded as torch's
        float *pp[3] = {&p0, &p1, &p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lv = a.base_lv[bi * a.lv_sb + c * a.lv_sc + nc * a.lv_sn];
            const float mu = a.base_mu[bi * a.mu_sb + c * a.mu_sc + nc * a.mu_sn];
            *pp[c] = __fadd_rn(__fmul_rn(*pp[c], expf(__fmul_rn(0.5f, lv))), mu);
        }
        if (a.z_out != nullptr && valid && !(lane >> 5)) {
            a.z_out[cloud + n] = p0; a.z_out[cloud + N + n] = p1; a.z_out[cloud + 2 * (size_t)N + n] = p2;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;   // running sum of logvar per channel
    constexpr bool inverse = INV;              // the direction is compiled in: the coupling update has no uniform branches
    const size_t list_stride = (size_t)a.B * 3 * N;

    float negone = -1.0f;                      // opaque to the compiler: see layer_pipe's fp16 split
    asm volatile("" : "+s"(negone));
    float zero_lv = 0.0f;                      // a kept channel's logvar, opaque so that its factor is computed, not folded
    asm volatile("" : "+s"(zero_lv));
    const float v_keep = a.eps + __expf(zero_lv);
    const float k_keep = inverse ? __builtin_amdgcn_rsqf(v_keep) : __builtin_amdgcn_sqrtf(v_keep);
    const int lfirst = inverse ? L - 1 : 0;
    auto stage_group = [&](int g) {                      // the LPB layers of steps g*LPB .. into buffer g & 1
#pragma unroll
        for (int k = 0; k < LPB; ++k) {
            const int st = g * LPB + k;
            if (st < L) stage_layer<NS, FW>(a, inverse ? L - 1 - st : st, bi, smem + ((g & 1) * LPB + k) * LBYTES, wave, lane);
        }
    };
    const bool lag = SKEW && wave >= FW / 2;
    auto stage_step = [&](int st) {                      // SKEW: the layer of step st into ring slot st % 3
        if ((DPF_ABLATE & 16) && st >= 3) return;        // timing experiment: no weight DMA after the first three layers
        if (st < L) stage_layer<NS, FW>(a, inverse ? L - 1 - st : st, bi, smem + (st % 3) * LBYTES, wave, lane);
    };
    if constexpr (SKEW) { stage_step(0); stage_step(1); } else stage_group(0);
    // Layer descriptors (keep/warp channels): lane l of every wave holds the rows of layers l and 64 + l, a
    // layer's row comes out with v_readlane.  (Loading them inside the loop puts a vector-memory wait at the top of every
    // layer, and vmcnt retires in order: it waited for the whole next-layer DMA issued just before -- r01
    // profile: ~3000 of the 9700 cycles per layer.)
    // ... packed into one word per layer: 2 bits per field (value + 1), so a layer's descriptor is ONE v_readlane and
    // four scalar bit-field extracts -- no branches (the four-readlane form compiled to a branch ladder per field)
    auto pack_meta = [&](int row) {
        const int4 m = ((const int4 *)a.meta)[min(row, L - 1)];
        return (m.x + 1) | ((m.y + 1) << 2) | ((m.z + 1) << 4) | ((m.w + 1) << 6);
    };
    const int code_lo = pack_meta(lane), code_hi = pack_meta(64 + lane);
    auto layer_meta = [&](int l, int &k0, int &k1, int &w0, int &w1) {      // L <= 128 (checked by the launcher)
        const int c = __builtin_amdgcn_readlane(l >= 64 ? code_hi : code_lo, l & 63);
        k0 = (c & 3) - 1; k1 = ((c >> 2) & 3) - 1; w0 = ((c >> 4) & 3) - 1; w1 = ((c >> 6) & 3) - 1;
    };
    int ka, kb, wa, wb;
    layer_meta(lfirst, ka, kb, wa, wb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first layers' weights (and everything above) have landed
    __syncthreads();

    for (int step = 0; step < L; ++step) {
        const int li = inverse ? L - 1 - step : step;
        const int ln = inverse ? (li > 0 ? li - 1 : 0) : (li + 1 < L ? li + 1 : li);
        const int grp = step / LPB, within = step - grp * LPB;
        const uint8_t *lb = smem + (SKEW ? step % 3 : (grp & 1) * LPB + within) * LBYTES;
        if constexpr (!SKEW)
            if (within == 0 && (grp + 1) * LPB < L) stage_group(grp + 1);
        int nka, nkb, nwa, nwb;
        layer_meta(ln, nka, nkb, nwa, nwb);
        unsigned long long tt[16];
        (void)tt;
        DPF_T(0)
        // ---- B operand of the input MFMA: 3-way bf16 split of this half's input channel
        const float xa = sel3(ka, p0, p1, p2);
        const float xb = kb < 0 ? 0.f : sel3(kb, p0, p1, p2);
        const float x = h ? xb : xa;
        const u32x4 b0 = input_fragment(x, h);

        float o[2][2];
        if (DPF_ABLATE & 128) {            // timing experiment: no conditioner at all (loop, staging, barriers, transform only)
            o[0][0] = p0 * 0.1f; o[0][1] = p1 * 0.1f; o[1][0] = p2 * 0.1f; o[1][1] = p0 * 0.05f;
            (void)b0;
        } else
        if constexpr (PIPE) {
            if (wb < 0) layer_pipe<NS, false, F16>(lb, lane, h, b0, negone, lag ? 5 : 0, o, tt);   // layer warps one channel
            else layer_pipe<NS, true, F16>(lb, lane, h, b0, negone, lag ? 5 : 0, o, tt);
        } else if (wb < 0) {                                // layer warps one channel
            branch_tile<NS, false>(lb, 0, lane, h, b0, o[0][0], o[0][1], tt);
            branch_tile<NS, false>(lb, 1, lane, h, b0, o[1][0], o[1][1], tt);
        } else {
            branch_tile<NS, true>(lb, 0, lane, h, b0, o[0][0], o[0][1], tt);
            branch_tile<NS, true>(lb, 1, lane, h, b0, o[1][0], o[1][1], tt);
        }
        DPF_T(4)
        const float *b2 = (const float *)(lb + FILMOFF) + FILM_B2_OFF;
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int e = 0; e < 2; ++e) o[br][e] = PIPE ? half_sum(o[br][e]) : half_sum(o[br][e]) + b2[br * 2 + e];   // layer_pipe starts its sums at b2 / 2
        // ---- coupling transform (flows.py:96-115); branch 0 = logvar, 1 = mu.  Only the warped channels (one or two,
        // wave-uniform) go through softsign / exp / sqrt; a kept channel has logvar 0, so its factor is the constant
        // sqrt(eps + exp(0)) (direct) or its reciprocal square root (inverse), formed once with the same instructions.
        float lva, lvb = 0.f, fa, fb = k_keep;
        if (DPF_ABLATE & 64) { lva = o[0][0]; lvb = o[0][1]; fa = a.eps + lva; fb = a.eps + lvb; }
        else {
            lva = o[0][0] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][0]));   // softsign, :99
            const float va = a.eps + __expf(lva);
            fa = inverse ? __builtin_amdgcn_rsqf(va) : __builtin_amdgcn_sqrtf(va);
            if (wb >= 0) {
                lvb = o[0][1] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][1]));
                const float vb = a.eps + __expf(lvb);
                fb = inverse ? __builtin_amdgcn_rsqf(vb) : __builtin_amdgcn_sqrtf(vb);
            }
        }
        float lv[3], mu[3], pn[3];
        const float pin[3] = {p0, p1, p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lv[c] = c == wa ? lva : (c == wb ? lvb : 0.f);
            mu[c] = c == wa ? o[1][0] : (c == wb ? o[1][1] : 0.f);
            const float f = c == wa ? fa : (c == wb ? fb : k_keep);
            // keep channels are scaled by sqrt(1 + eps) too, as in the reference (:113/:115)
            pn[c] = inverse ? (pin[c] - mu[c]) * f : f * pin[c] + mu[c];
        }
        p0 = pn[0]; p1 = pn[1]; p2 = pn[2];
        s0 += lv[0]; s1 += lv[1]; s2 += lv[2];
        // r03: the ring's protocol says "the DMA of layer n + 1, issued after barrier n - 1, was waited for by its issuers
        // before they arrive at barrier n" -- but nothing in the compiled loop did: __syncthreads() emits no vmcnt wait on
        // gfx950 (back-off barrier) and the compiler's LDS-DMA alias wait did not appear in this loop (ISA: no `vmcnt` between
        // the loop's top and bottom), so a landed DMA was a matter of timing (a layer takes ~3 us, the DMA ~1).  Explicit now,
        // and placed BEFORE this layer's list stores so that it waits only for operations issued a layer ago (free).
        if constexpr (SKEW) {
            if (!lag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (a.ps != nullptr && valid) {   // per-layer lists in DIRECT order (decoders.py:61-70); halves share the rows
            const size_t base = (size_t)li * list_stride + cloud + n;
            float *dst[5]; float val[5];
            dst[0] = (h ? a.mus + base + 2 * (size_t)N : a.ps + base);              val[0] = h ? mu[2] : pn[0];
            dst[1] = (h ? a.lvs + base : a.ps + base + N);                          val[1] = h ? lv[0] : pn[1];
            dst[2] = (h ? a.lvs + base + N : a.ps + base + 2 * (size_t)N);          val[2] = h ? lv[1] : pn[2];
            dst[3] = (h ? a.lvs + base + 2 * (size_t)N : a.mus + base);             val[3] = h ? lv[2] : mu[0];
            dst[4] = a.mus + base + N;                                             val[4] = mu[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) *dst[e] = val[e];
            if (!h) *dst[4] = val[4];
        }
        ka = nka; kb = nkb; wa = nwa; wb = nwb;
        DPF_T(5)
#ifdef DPF_PROFILE
        if (within == LPB - 1 && !lag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA pieces landed (split from the barrier wait)
#endif
        DPF_T(6)
        if constexpr (SKEW) {
            if (!lag) {               // leaders: barrier n; layer n - 1's slot is free now -> layer n + 2
                if (!(DPF_ABLATE & 32)) __syncthreads();
                stage_step(step + 2);
            }
        } else {
            if (within == LPB - 1) {   // the next buffer's weights have landed (explicitly: see above); everyone is done with this one
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
#ifdef DPF_PROFILE
        DPF_T(7)
        if (a.prof != nullptr && lane == 0 && blockIdx.x < 2 && blockIdx.y == 0) {
            unsigned long long *o2 = a.prof + (((size_t)(blockIdx.x * FW + wave)) * L + step) * 16;
            for (int i = 0; i < 16; ++i) o2[i] = tt[i];
        }
#endif
    }
    if (valid) {
        if (!h) {
            a.p_out[cloud + n] = p0; a.p_out[cloud + N + n] = p1; a.p_out[cloud + 2 * (size_t)N + n] = p2;
            if (a.p_out_pm != nullptr) {   // point-major (B,N,3) copy for the structural losses (evaluating.py:110)
                float *o2 = a.p_out_pm + ((size_t)bi * N + n) * 3;
                o2[0] = p0; o2[1] = p1; o2[2] = p2;
            }
        } else if (a.sum_lv != nullptr) {
            a.sum_lv[cloud + n] = s0; a.sum_lv[cloud + N + n] = s1; a.sum_lv[cloud + 2 * (size_t)N + n] = s2;
        }
    }
    if (a.xs_part != nullptr) {   // sum xa, xb, xa^2, xb^2, xa*xb over this workgroup's points, fixed order, as doubles
        __syncthreads();                               // everybody is done with the weight buffers
        double *lanes = (double *)smem, *red = lanes + FW * 5 * 32;       // [wave][moment][point], [wave][moment]
        if (!h) {
            double xa = 0, xb = 0;
            if (valid) { xa = sel3(a.xs_ka, p0, p1, p2); xb = a.xs_kb >= 0 ? sel3(a.xs_kb, p0, p1, p2) : 0.f; }
            double *q = lanes + wave * 5 * 32 + pl;
            q[0] = xa; q[32] = xb; q[64] = xa * xa; q[96] = xb * xb; q[128] = xa * xb;
        }
        __syncthreads();
        if (threadIdx.x < FW * 5) {                    // thread = (wave, moment): 32 points, four accumulators
            const double *q = lanes + threadIdx.x * 32;
            double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
            for (int i = 0; i < 32; i += 4) { t0 += q[i]; t1 += q[i + 1]; t2 += q[i + 2]; t3 += q[i + 3]; }
            red[threadIdx.x] = (t0 + t1) + (t2 + t3);
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            double t = 0;
            for (int w = 0; w < FW; ++w) t += red[w * 5 + threadIdx.x];
            a.xs_part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + threadIdx.x] = t;
        }
    }
}

#ifdef DPF_PROFILE
unsigned long long *g_prof = nullptr;
#endif
int ns_of(int precision) {   // number of operand parts; DPF_PREC_F16X3 shares the hi/lo layout of bf16x3
    return precision == DPF_PREC_BF16 ? 1 : (precision == DPF_PREC_BF16X3 || precision == DPF_PREC_F16X3) ? 2
           : precision == DPF_PREC_BF16X6 ? 3 : 0;
}

template <int NS, int FW, int LPB, bool PIPE, bool F16, bool SKEW, bool INV>
int launch_flow_dir(const FlowArgs &a, hipStream_t s) {
    const int lds = (SKEW ? 3 : 2 * LPB) * (p_layer_bytes(NS) + FILM_BYTES);
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)flow_kernel<NS, FW, LPB, PIPE, F16, SKEW, INV>, lds); e != hipSuccess) return (int)e;
    const dim3 grid((a.N + TILE * FW - 1) / (TILE * FW), a.B), block(FW * 64);
    hipLaunchKernelGGL((flow_kernel<NS, FW, LPB, PIPE, F16, SKEW, INV>), grid, block, lds, s, a);
    return (int)hipGetLastError();
}
template <int NS, int FW, int LPB, bool PIPE, bool F16, bool SKEW = false>
int launch_flow(const FlowArgs &a, hipStream_t s) {
    return a.mode == DPF_MODE_INVERSE ? launch_flow_dir<NS, FW, LPB, PIPE, F16, SKEW, true>(a, s)
                                      : launch_flow_dir<NS, FW, LPB, PIPE, F16, SKEW, false>(a, s);
}
// bf16 / bf16x3 / f16x3 run the pipelined layer body (layer_pipe); bf16x6 keeps one branch at a time (branch_tile:
// three operand parts per branch do not fit two branches' fragments into the 256 VGPRs of two waves per SIMD)
template <int FW, int LPB>
int launch_flow_prec(int precision, const FlowArgs &a, hipStream_t s) {
    if constexpr (FW == 8 && LPB == 0) {          // 8-wave workgroups of the two-part precisions: the skewed ring
        switch (precision) {
            case DPF_PREC_BF16: return launch_flow<1, 8, 1, true, false, true>(a, s);
            case DPF_PREC_BF16X3: return launch_flow<2, 8, 1, true, false, true>(a, s);
            case DPF_PREC_F16X3: return launch_flow<2, 8, 1, true, true, true>(a, s);
            default: return launch_flow<3, 8, 1, false, false>(a, s);
        }
    }
    constexpr int LP = LPB ? LPB : 1;
    switch (precision) {
        case DPF_PREC_BF16: return launch_flow<1, FW, LP, true, false>(a, s);
        case DPF_PREC_BF16X3: return launch_flow<2, FW, LP, true, false>(a, s);
        case DPF_PREC_F16X3: return launch_flow<2, FW, LP, true, true>(a, s);
        default: return launch_flow<3, FW, 1, false, false>(a, s);
    }
}

}  // namespace

extern "C" size_t dpf_flow_canon_floats(int G) { return (size_t)c_layer_floats(G); }

// bytes of the MFMA-fragment weights in front of the conditioner weights: the 32-point-tile layout of every layer, then
// (f16x3 only) the 16-point-tile layout of every layer (csrc/flow16.hip)
static size_t packed_frag_bytes(int n_layers, int precision) {
    return (size_t)n_layers * (p_layer_bytes(ns_of(precision)) + (precision == DPF_PREC_F16X3 ? P16_LAYER : 0));
}

extern "C" size_t dpf_flow_packed_bytes(int n_layers, int G, int precision) {
    const int ns = ns_of(precision);
    return ns ? packed_frag_bytes(n_layers, precision) + fw_total_floats(n_layers, G) * sizeof(float) : 0;
}

extern "C" size_t dpf_flow_film_floats(int n_layers, int B) { return (size_t)n_layers * B * (FILM_BYTES / 4); }

extern "C" int dpf_flow_pack(int n_layers, int G, int precision, const float *canon, const int *meta, void *packed,
                             dpf_stream_t stream) {
    (void)meta;
    const int ns = ns_of(precision);
    if (!ns || n_layers < 0 || G <= 0) return DPF_EINVAL;
    if (n_layers == 0) return 0;
    if (!canon || !packed) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (ns == 1) hipLaunchKernelGGL(pack_kernel<1>, dim3(n_layers), dim3(256), 0, s, G, canon, (uint8_t *)packed);
    if (ns == 2 && precision != DPF_PREC_F16X3)
        hipLaunchKernelGGL(pack_kernel<2>, dim3(n_layers), dim3(256), 0, s, G, canon, (uint8_t *)packed);
    if (precision == DPF_PREC_F16X3)
        hipLaunchKernelGGL((pack_kernel<2, true>), dim3(n_layers), dim3(256), 0, s, G, canon, (uint8_t *)packed);
    if (ns == 3) hipLaunchKernelGGL(pack_kernel<3>, dim3(n_layers), dim3(256), 0, s, G, canon, (uint8_t *)packed);
    if (precision == DPF_PREC_F16X3)
        if (int rc = flow16_pack(n_layers, G, canon, (uint8_t *)packed + (size_t)n_layers * p_layer_bytes(ns), s)) return rc;
    float *fw = (float *)((uint8_t *)packed + packed_frag_bytes(n_layers, precision));
    hipLaunchKernelGGL(pack_film_kernel, dim3(n_layers * 4), dim3(256), 0, s, n_layers, G, canon, fw, precision == DPF_PREC_F16X3 ? 1 : 0);
    return (int)hipGetLastError();
}

extern "C" int dpf_flow_film(int n_layers, int B, int G, int precision, const void *packed, const float *g,
                             float *film, float flow_eps, dpf_stream_t stream) {
    const int ns = ns_of(precision);
    if (!ns || n_layers < 0 || B < 0 || G <= 0) return DPF_EINVAL;
    if (n_layers == 0 || B == 0) return 0;
    if (!packed || !g || !film) return DPF_EINVAL;
    if (G % 16 != 0 || G > 2048) return DPF_ENOSUP;
    const float *fw = (const float *)((const uint8_t *)packed + packed_frag_bytes(n_layers, precision));
    const int lds = (FILM_CLOUDS * G + (2 * 4 + 2 + 1) * FILM_CLOUDS * 64) * (int)sizeof(float);
    static LdsLimit film_limit;
    if (lds > 65536)
        if (hipError_t e = film_limit.ensure((const void *)film_kernel, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(film_kernel, dim3(n_layers * 2, (B + FILM_CLOUDS - 1) / FILM_CLOUDS), dim3(512), lds,
                       (hipStream_t)stream, n_layers, B, G, fw, g, film, flow_eps);
    return (int)hipGetLastError();
}

static int flow_forward_impl(int n_layers, int B, int N, int mode, int precision, const void *packed,
                             const int *meta, const float *film, const float *p_in, float *p_out,
                             float *p_out_pointmajor, float *sum_logvar, float *ps, float *mus, float *logvars,
                             float flow_eps, dpf_stream_t stream, const float *base_mu, const long *mu_strides,
                             const float *base_lv, const long *lv_strides, float *z_out, double *xs_part = nullptr,
                             int xs_ka = 0, int xs_kb = -1, int *xs_rows = nullptr, bool packed16_ok = true) {
    const int ns = ns_of(precision);
    if (!ns || n_layers <= 0 || B < 0 || N <= 0 || (mode != DPF_MODE_DIRECT && mode != DPF_MODE_INVERSE)) return DPF_EINVAL;
    if (B == 0) return 0;
    if (!packed || !meta || !film || !p_in || !p_out) return DPF_EINVAL;
    if ((ps != nullptr) != (mus != nullptr) || (ps != nullptr) != (logvars != nullptr)) return DPF_EINVAL;
    if (B > 65535 || n_layers > 128) return DPF_ENOSUP;
    FlowArgs a;
    a.packed = (const uint8_t *)packed; a.meta = meta; a.film = film; a.p_in = p_in;
    a.p_out = p_out; a.p_out_pm = p_out_pointmajor; a.sum_lv = sum_logvar; a.ps = ps; a.mus = mus; a.lvs = logvars;
    a.L = n_layers; a.B = B; a.N = N; a.mode = mode; a.eps = flow_eps;
    a.base_mu = base_mu; a.base_lv = base_lv; a.z_out = z_out;
    a.xs_part = xs_part; a.xs_ka = xs_ka; a.xs_kb = xs_kb;
    a.mu_sb = a.mu_sc = a.mu_sn = a.lv_sb = a.lv_sc = a.lv_sn = 0;
    a.prof = nullptr;
#ifdef DPF_PROFILE
    a.prof = g_prof;
#endif
    if (base_mu != nullptr) {
        a.mu_sb = mu_strides[0]; a.mu_sc = mu_strides[1]; a.mu_sn = mu_strides[2];
        a.lv_sb = lv_strides[0]; a.lv_sc = lv_strides[1]; a.lv_sn = lv_strides[2];
    }
    hipStream_t s = (hipStream_t)stream;
    // small batches (at most one 16-point tile per SIMD of the chip): the 16-point-tile kernel, csrc/flow16.hip
    a.packed16 = nullptr;
    if (packed16_ok && flow16_serves(n_layers, B, N, precision, xs_part != nullptr)) {
        a.packed16 = (const uint8_t *)packed + (size_t)n_layers * p_layer_bytes(ns);
        return flow16_launch(&a, s);
    }
    // 8-wave workgroups (256 points of one cloud) unless that leaves CUs without a workgroup
    static const int force_fw = getenv("DPF_FLOW_WAVES") ? atoi(getenv("DPF_FLOW_WAVES")) : 0;
    // waves (32-point tiles) per workgroup: as many as possible (each workgroup streams the layer
    // weights through its own LDS) while the launch still has a workgroup for every CU
    // (measured r01: below 4 waves the LDS-DMA fill of a layer, ~1.5 us for 38 KB on one CU, is no
    // longer hidden, so 2- and 1-wave workgroups are only for clouds of <= 64 / <= 32 points)
    int fw = force_fw;
    if (!fw) {
        fw = 8;
        if ((long)B * ((N + 255) / 256) < 224) fw = 4;
        if (N <= 64) fw = 2;
        if (N <= 32) fw = 1;
    }
    if (xs_rows != nullptr) *xs_rows = (N + TILE * (fw >= 8 ? 8 : fw >= 4 ? 4 : fw >= 2 ? 2 : 1) - 1) / (TILE * (fw >= 8 ? 8 : fw >= 4 ? 4 : fw >= 2 ? 2 : 1));
    // two layers per LDS buffer where a CU gets one workgroup anyway and the 2 x 2 layers fit its LDS (two-part precisions)
    static const int lpb_env = getenv("DPF_FLOW_LPB") ? atoi(getenv("DPF_FLOW_LPB")) : 0;
    const bool pair_ok = ns <= 2 && n_layers >= 2 && lpb_env != 1 && (lpb_env == 2 || (long)B * ((N + 255) / 256) <= 256);
    static const int skew_env = getenv("DPF_FLOW_SKEW") ? atoi(getenv("DPF_FLOW_SKEW")) : 1;
    if (fw >= 8 && skew_env && ns <= 2) return launch_flow_prec<8, 0>(precision, a, s);
    if (fw >= 8) return pair_ok ? launch_flow_prec<8, 2>(precision, a, s) : launch_flow_prec<8, 1>(precision, a, s);
    if (fw >= 4) return launch_flow_prec<4, 1>(precision, a, s);
    if (fw >= 2) return launch_flow_prec<2, 1>(precision, a, s);
    return launch_flow_prec<1, 1>(precision, a, s);
}

extern "C" int dpf_flow_forward(int n_layers, int B, int N, int mode, int precision, const void *packed,
                                const int *meta, const float *film, const float *p_in, float *p_out,
                                float *p_out_pointmajor, float *sum_logvar, float *ps, float *mus, float *logvars,
                                float flow_eps, dpf_stream_t stream) {
    return flow_forward_impl(n_layers, B, N, mode, precision, packed, meta, film, p_in, p_out, p_out_pointmajor, sum_logvar, ps,
                             mus, logvars, flow_eps, stream, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// csrc-internal (flow_train.hip): one training-mode layer through dpf_flow_forward that also leaves the moments of the
// output's channels ka / kb per workgroup in xs_part (tstats_x_kernel's layout, *xs_rows partial rows per cloud)
int flow_forward_xstats(int B, int N, int mode, int precision, const void *packed, const int *meta, const float *film,
                        const float *p_in, float *ps, float *mus, float *logvars, float flow_eps, dpf_stream_t stream,
                        double *xs_part, int xs_ka, int xs_kb, int *xs_rows) {
    return flow_forward_impl(1, B, N, mode, precision, packed, meta, film, p_in, ps, nullptr, nullptr, ps, mus, logvars, flow_eps,
                             stream, nullptr, nullptr, nullptr, nullptr, nullptr, xs_part, xs_ka, xs_kb, xs_rows,
                             /*packed16_ok=*/false);    // this block was packed by flow_train.hip's tpack_kernel: 32-point layout only
}

// dpf_flow_forward in DIRECT mode with the reparameterisation of the base sample fused into its prologue
// (models.py:76-79, :212): noise (B,3,N) in, z = noise * exp(0.5 * lv0) + mu0 formed in registers (and stored to z_out,
// which the models return as p_prior_samples[0]); mu0 / lv0 through (batch, channel, point) element strides.
extern "C" int dpf_flow_forward_base(int n_layers, int B, int N, int precision, const void *packed, const int *meta,
                                     const float *film, const float *noise, const float *mu0, long mu_sb, long mu_sc,
                                     long mu_sn, const float *lv0, long lv_sb, long lv_sc, long lv_sn, float *z_out,
                                     float *p_out, float *p_out_pointmajor, float *sum_logvar, float *ps, float *mus,
                                     float *logvars, float flow_eps, dpf_stream_t stream) {
    if (!mu0 || !lv0) return DPF_EINVAL;
    const long ms[3] = {mu_sb, mu_sc, mu_sn}, ls[3] = {lv_sb, lv_sc, lv_sn};
    return flow_forward_impl(n_layers, B, N, DPF_MODE_DIRECT, precision, packed, meta, film, noise, p_out, p_out_pointmajor,
                             sum_logvar, ps, mus, logvars, flow_eps, stream, mu0, ms, lv0, ls, z_out);
}

#ifdef DPF_PROFILE
extern "C" void dpf_debug_set_prof(void *p) { g_prof = (unsigned long long *)p; }
#endif
extern "C" const char *dpf_version(void) { return "dpf_hip gfx950 r1"; }
