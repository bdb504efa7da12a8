// Fused PointNet cloud encoder + max over the points for gfx950 (MI355X), eval-mode BatchNorm.
//
// Replaces
//   PointNetCloudEncoder.forward           lib/networks/encoders.py:27-28
//     features = [SharedDot(no bias) . BatchNorm1d . ReLU] x 4,  3 -> 64 -> 128 -> 256 -> 512   (encoders.py:15-25)
//   torch.max(features, dim=2)[0]          lib/networks/models.py:85,131,175
// (12 ATen kernels writing and re-reading (B,C,N) activations: (64+128+256+512) * 4 B = 3.8 KB per point per pass,
// ~1 GB of HBM traffic at B=32, N=2048) by ONE kernel that reads 12 B per point and writes B*512 floats: the
// activations never leave the register file.
//
// A wave owns a tile of 32 points; a workgroup = 8 waves = 256 points of one cloud.  Per layer the weights are the
// MFMA A operand (rows = output features) and the points the B operand, so the accumulator fragment of one
// v_mfma_f32_32x32x16_bf16 -- each lane holds 16 features of ITS point -- becomes, after ReLU and a bf16 hi/lo split
// in registers, the B fragment of the next layer (the K-slot <-> feature permutation this implies is applied to the
// next layer's columns at pack time; same trick as csrc/flow.hip).  The LAST layer swaps the operands (activations
// = A, weights = B): then a lane holds 16 POINTS of its feature and the max over the tile is 8 v_max3 in-lane plus
// one cross-half swap, instead of a 5-step cross-lane butterfly per accumulator register.
//
// Split precision (bf16x3 default): products hi*hi + hi*lo + lo*hi of the bf16 hi/lo parts, fp32 accumulate.
// BatchNorm (running statistics) is folded: scale into the weight rows before the split, shift as the accumulator's
// initial value (layer 0: a constant-1 K slot of the input MFMA, whose 16 K slots hold the 3-way split of x, y, z).
//
// The 672 KiB (bf16x3) of layer 1-3 fragments stream through LDS in 11 chunks of 64 KiB (double-buffered,
// global_load_lds, one workgroup barrier per chunk).
#include "flow_common.h"
#include "encoder_layout.h"
#include "zero_fill.h"

namespace {

// packed: [A0 4 KiB: [t2][ks2][lane64][8] | bias 4 KiB: b1acc[4][2][16] b2acc[8][2][16] b3[512] pad | chunks]
// The fragments of layers 1-3 form one stream of 336 slots (a slot = the NS parts of one 1 KiB fragment):
// layer 1: 4 M tiles x 4 k-steps at slot 0, layer 2: 8 x 8 at slot 16, layer 3: 16 N tiles x 16 at slot 80; a tile's
// k-steps are consecutive and never straddle a chunk of e_slots(NS) slots.  Chunk: [part][slot][lane64][8].
constexpr int EP_A0 = 0, EP_BIAS = 4096, EP_CHUNKS = 8192;
constexpr int EB_1 = 0, EB_2 = 128, EB_3 = 384;                      // float offsets inside the bias block
constexpr int ES_L1 = 0, ES_L2 = 16, ES_L3 = 80, ES_TOTAL = 336;
// slots per chunk: 32 (two 64 KiB buffers at bf16x3) halves the number of workgroup barriers; bf16x6 keeps 16
__host__ __device__ constexpr int e_slots(int NS) { return NS == 3 ? 16 : 32; }
__host__ __device__ constexpr int e_nchunk(int NS) { return (ES_TOTAL + e_slots(NS) - 1) / e_slots(NS); }
__host__ __device__ constexpr int ep_chunk_bytes(int NS) { return NS * e_slots(NS) * 1024; }
__host__ __device__ constexpr size_t ep_bytes(int NS) { return EP_CHUNKS + (size_t)e_nchunk(NS) * ep_chunk_bytes(NS); }

// A workgroup covers 8 tiles (256 points) of one cloud.  TP = tiles per wave:
//   TP = 1 (default): 8 waves, two per SIMD with 256 VGPRs each (bf16x6: 4 waves, one per SIMD -- it keeps 192 VGPRs
//          of layer-3 operand fragments);
//   TP = 2 (-DDPF_ENC_TP=2; bf16, bf16x3): 4 waves, one per SIMD with 484 VGPRs; every weight fragment read from
//          LDS feeds the MFMAs of both tiles, halving the LDS->VGPR traffic.  Measured r01 at cfg-2: 56.1 us vs
//          50.5 us for TP = 1 -- a lone wave per SIMD does not hide its own LDS and MFMA latencies.
#ifndef DPF_ENC_TP
#define DPF_ENC_TP 1
#endif
__host__ __device__ constexpr int e_tp(int NS) { return NS == 3 ? 1 : DPF_ENC_TP; }
__host__ __device__ constexpr int e_waves(int NS) { return NS == 3 ? 4 : 8 / e_tp(NS); }

// K index held by element j of lane-half kg in k-step ks = the feature that register 8*(ks&1)+j of accumulator
// tile ks>>1 holds in lane-half kg (acc_feature)
__host__ __device__ constexpr int k_feature(int ks, int j, int kg) { return acc_feature(ks >> 1, 8 * (ks & 1) + j, kg); }

template <int NS>
__global__ __launch_bounds__(256) void enc_pack_kernel(const float *__restrict__ canon, uint8_t *__restrict__ packed) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    auto scale_shift = [&](int l, int f, float &s, float &t) {
        const float *c = canon + e_layer_off(l) + e_cout(l) * e_cin(l);
        const int co = e_cout(l);
        s = c[f] / sqrtf(c[3 * co + f] + BN_EPS);
        t = c[co + f] - c[2 * co + f] * s;
    };
    // A0: the three input channels over two k-steps (x, y + shift | z)
    uint16_t *a0 = (uint16_t *)(packed + EP_A0);
    for (int idx = tid; idx < 2 * 2 * 64 * 8; idx += nth) {
        const int j = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 1, t = idx >> 10;
        const int f = 32 * t + (lane & 31), h = lane >> 5;
        float s, sh;
        scale_shift(0, f, s, sh);
        const float *w = canon + e_layer_off(0) + f * EC0;
        uint32_t v;
        if (ks == 0) v = input_weight_slot(s * w[h], sh, h, j);
        else v = h == 0 ? input_weight_slot(s * w[2], 0.f, 0, j) : 0u;
        a0[idx] = (uint16_t)v;
    }
    float *bias = (float *)(packed + EP_BIAS);
    for (int idx = tid; idx < 1024; idx += nth) {
        float s, sh = 0.f;
        if (idx < EB_2) {                       // layer 1, accumulator order [mt4][h2][r16]
            scale_shift(1, acc_feature(idx >> 5, idx & 15, (idx >> 4) & 1), s, sh);
        } else if (idx < EB_3) {
            const int i = idx - EB_2;
            scale_shift(2, acc_feature(i >> 5, i & 15, (i >> 4) & 1), s, sh);
        } else if (idx < EB_3 + EC4) {
            scale_shift(3, idx - EB_3, s, sh);
        }
        bias[idx] = sh;
    }
    uint16_t *ch = (uint16_t *)(packed + EP_CHUNKS);
    constexpr int S = e_slots(NS);
    const int per_chunk = S * 64 * 8;           // elements per part
    for (int idx = tid; idx < e_nchunk(NS) * per_chunk; idx += nth) {
        const int c = idx / per_chunk, e = idx % per_chunk;
        const int j = e & 7, lane = (e >> 3) & 63, slot = c * S + (e >> 9);
        uint16_t *o = ch + (size_t)c * NS * per_chunk + e;
        if (slot >= ES_TOTAL) {                 // padding of the last chunk
            for (int part = 0; part < NS; ++part) o[part * per_chunk] = 0;
            continue;
        }
        int l, rt, ks;
        if (slot < ES_L2) { l = 1; rt = slot >> 2; ks = slot & 3; }
        else if (slot < ES_L3) { l = 2; rt = (slot - ES_L2) >> 3; ks = (slot - ES_L2) & 7; }
        else { l = 3; rt = (slot - ES_L3) >> 4; ks = (slot - ES_L3) & 15; }
        const int row = 32 * rt + (lane & 31), col = k_feature(ks, j, lane >> 5);
        float s, sh;
        scale_shift(l, row, s, sh);
        const float w = s * canon[e_layer_off(l) + row * e_cin(l) + col];
        if (NS == 1) {
            o[0] = (uint16_t)bf16_rne(w);
        } else if (NS == 2) {
            float r1;
            o[0] = (uint16_t)(split_hi(w, r1) >> 16);
            o[per_chunk] = (uint16_t)bf16_rne(r1);
        } else {
            float r1, r2;
            o[0] = (uint16_t)(split_hi(w, r1) >> 16);
            o[per_chunk] = (uint16_t)(split_hi(r1, r2) >> 16);
            o[2 * per_chunk] = (uint16_t)bf16_rne(r2);
        }
    }
}

struct EncArgs {
    const uint8_t *packed;
    const float *x;        // (B,3,N)
    float *gmax;           // (B,512), zero-initialised by the launcher; updated with integer atomicMax (values >= 0)
    float *feat;           // (B,512,N) or NULL
    int B, N;
};

// relu + split of one accumulator tile into the two k-steps 2t, 2t+1 of the next layer's fragments
template <int NS>
__device__ __forceinline__ void relu_split(const f32x16 &acc, u32x4 (&dst)[NS][16], int t) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float v0 = relu(acc[r]), v1 = relu(acc[r + 1]);
        const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;
        if (NS == 1) {
            dst[0][s][d] = pack_bf16_rne(v0, v1);
        } else if (NS == 2) {
            float l0, l1;
            split_hi(v0, l0); split_hi(v1, l1);
            dst[0][s][d] = pack_bf16_trunc(v0, v1);
            dst[1][s][d] = pack_bf16_rne(l0, l1);
        } else {
            float l0, l1, m0, m1;
            split_hi(v0, l0); split_hi(v1, l1);
            split_hi(l0, m0); split_hi(l1, m1);
            dst[0][s][d] = pack_bf16_trunc(v0, v1);
            dst[1][s][d] = pack_bf16_trunc(l0, l1);
            dst[2][s][d] = pack_bf16_rne(m0, m1);
        }
    }
}

template <int NS>
__device__ __forceinline__ void stage_chunk(const uint8_t *packed, int c, uint8_t *lds, int wave, int lane) {
    constexpr int NI = ep_chunk_bytes(NS) / 1024, EW = e_waves(NS);     // wave-instructions of 1 KiB
    const uint8_t *src = packed + EP_CHUNKS + (size_t)c * ep_chunk_bytes(NS);
#pragma unroll
    for (int i = 0; i < NI / EW; ++i) {
        const int k = wave + i * EW;
        __builtin_amdgcn_global_load_lds((glb_void *)(src + k * 1024 + lane * 16), (lds_void *)(lds + k * 1024), 16, 0, 0);
    }
}

// One output tile for each of the wave's TP point tiles: K k-steps of fragments at slots [slot0, slot0 + K) of the
// chunk at cb; every fragment read feeds the MFMAs of all TP tiles.  SWAP: the activations are the A operand and the
// weights the B operand.  NA accumulators per tile (even / odd k-steps) keep consecutive MFMAs independent.
template <int NS, int K, bool SWAP, int TP, int NA>
__device__ __forceinline__ void tile_gemm(const uint8_t *cb, int slot0, int lane, const u32x4 (&act)[TP][NS][16],
                                          f32x16 (&out)[TP]) {     // out: in = initial value, out = result
    typedef Terms<NS> TT;
    f32x16 acc[TP][NA];
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        acc[q][0] = out[q];
        if (NA == 2) acc[q][NA - 1] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    u32x4 wf[2][NS];
    auto load = [&](int ks, u32x4 (&dst)[NS]) {
#pragma unroll
        for (int part = 0; part < NS; ++part)
            dst[part] = *(const u32x4 *)(cb + part * (e_slots(NS) * 1024) + ((slot0 + ks) * 64 + lane) * 16);
    };
    load(0, wf[0]);
#pragma unroll
    for (int ks = 0; ks < K; ++ks) {
        if (ks + 1 < K) load(ks + 1, wf[(ks + 1) & 1]);
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int q = 0; q < TP; ++q) {
                const u32x4 w = wf[ks & 1][TT::A[term]], x = act[q][TT::B[term]][ks];
                f32x16 &d = acc[q][ks & (NA - 1)];
                d = SWAP ? mfma(x, w, d) : mfma(w, x, d);
            }
    }
    // (element by element through an opaque copy: a vector `+` becomes v_pk_add_f32, which the scheduler then places directly in
    // front of the next tile's first MFMA -- the one pairing tools/asm_bisect found losing a packed result in csrc/emd.hip's
    // vectorised build; tools/mfma_overlap_check.py --no-packed-before-mfma gates every object on it)
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        if (NA == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float hi = acc[q][NA - 1][r];
                asm volatile("" : "+v"(hi));
                out[q][r] = acc[q][0][r] + hi;
            }
        } else {
            out[q] = acc[q][0];
        }
    }
}

__device__ __forceinline__ float half_max(float x) {   // max(x(lane), x(lane ^ 32))
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    return fmaxf(u2f(r[0]), u2f(r[1]));
}

template <int NS>
__global__ __launch_bounds__(e_waves(NS) * 64) void enc_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int CHB = ep_chunk_bytes(NS), EW = e_waves(NS), TP = e_tp(NS);
    // accumulators per tile and layer: two where the registers allow it (one wave per SIMD has nobody else to fill
    // the gap between dependent MFMAs)
    constexpr int NA12 = (EW == 4 && TP == 1) ? 2 : 1, NA3 = TP == 2 ? 1 : 2;
    uint8_t *l_a0 = smem, *l_bias = smem + 4096, *l_buf = smem + 8192;
    float *l_wmax = (float *)(smem + 8192 + 2 * CHB);             // [EW][512]

    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    const float *xc = a.x + (size_t)bi * 3 * N;
    int tile0[TP];                                                // first point of each of this wave's tiles
    float px[TP], py[TP], pz[TP];
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        tile0[q] = ((blockIdx.x * EW + wave) * TP + q) * TILE;
        const int nc = min(tile0[q] + pl, N - 1);
        px[q] = xc[nc]; py[q] = xc[N + nc]; pz[q] = xc[2 * (size_t)N + nc];
    }

    // A0 + bias block (8 KiB) and chunk 0
#pragma unroll
    for (int k = wave; k < 8; k += EW)
        __builtin_amdgcn_global_load_lds((glb_void *)(a.packed + k * 1024 + lane * 16), (lds_void *)(smem + k * 1024), 16, 0, 0);
    stage_chunk<NS>(a.packed, 0, l_buf, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces have landed (explicit: __syncthreads() alone emits no vmcnt wait on gfx950)
    __syncthreads();

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- layer 0: 3 -> 64 on the matrix core, fp32-accurate (3-way split of x, y | z)
    u32x4 f1[TP][NS][16];        // only k-steps 0..3 are used
#pragma unroll
    for (int q = 0; q < TP; ++q) {
        const u32x4 b0 = input_fragment(h ? py[q] : px[q], h);
        u32x4 b1 = input_fragment(pz[q], 0);
        if (h) b1 = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 a00 = *(const u32x4 *)(l_a0 + ((t * 2 + 0) * 64 + lane) * 16);
            const u32x4 a01 = *(const u32x4 *)(l_a0 + ((t * 2 + 1) * 64 + lane) * 16);
            f32x16 acc = mfma(a00, b0, zero16);
            acc = mfma(a01, b1, acc);
            relu_split<NS>(acc, f1[q], t);
        }
    }
    auto bias_tile = [&](int off, int mt) {       // accumulator-order shift of M tile mt
        const float *bp = (const float *)l_bias + off + (mt * 2 + h) * 16;
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4 *)(bp + 4 * q);
            v[4 * q + 0] = b.x; v[4 * q + 1] = b.y; v[4 * q + 2] = b.z; v[4 * q + 3] = b.w;
        }
        return v;
    };
    // Chunk `cur` is resident in buffer cur & 1 and chunk cur + 1 is on its way into the other one.  Moving on to the
    // next chunk is one barrier (it has landed; everybody is done with the buffer the one after it will overwrite).
    constexpr int S = e_slots(NS), NCH = e_nchunk(NS);
    int cur = 0;
    stage_chunk<NS>(a.packed, 1, l_buf + CHB, wave, lane);
    auto chunk_of = [&](int slot) -> const uint8_t * {       // slot: wave-uniform
        const int c = slot / S;
        if (c != cur) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk c's pieces (issued a chunk ago) have landed
            __syncthreads();
            cur = c;
            if (c + 1 < NCH) stage_chunk<NS>(a.packed, c + 1, l_buf + ((c + 1) & 1) * CHB, wave, lane);
        }
        return l_buf + (c & 1) * CHB;
    };
    // ---- layer 1: 64 -> 128
    u32x4 f2[TP][NS][16];        // k-steps 0..7
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int slot = ES_L1 + 4 * mt;
        const uint8_t *cb = chunk_of(slot);
        f32x16 acc[TP];
#pragma unroll
        for (int q = 0; q < TP; ++q) acc[q] = bias_tile(EB_1, mt);
        tile_gemm<NS, 4, false, TP, NA12>(cb, slot % S, lane, f1, acc);
#pragma unroll
        for (int q = 0; q < TP; ++q) relu_split<NS>(acc[q], f2[q], mt);
    }
    // ---- layer 2: 128 -> 256
    u32x4 f3[TP][NS][16];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int slot = ES_L2 + 8 * mt;
        const uint8_t *cb = chunk_of(slot);
        f32x16 acc[TP];
#pragma unroll
        for (int q = 0; q < TP; ++q) acc[q] = bias_tile(EB_2, mt);
        tile_gemm<NS, 8, false, TP, NA12>(cb, slot % S, lane, f2, acc);
#pragma unroll
        for (int q = 0; q < TP; ++q) relu_split<NS>(acc[q], f3[q], mt);
    }
    // ---- layer 3: 256 -> 512, operands swapped: accumulator register r = point (r&3) + 8*(r>>2) + 4h of the tile,
    //      lane column = output feature; one 32-feature tile at a time.
    for (int nt = 0; nt < 16; ++nt) {
        const int slot = ES_L3 + 16 * nt;
        const uint8_t *cb = chunk_of(slot);
        f32x16 acc[TP];
#pragma unroll
        for (int q = 0; q < TP; ++q) acc[q] = zero16;
        tile_gemm<NS, 16, true, TP, NA3>(cb, slot % S, lane, f3, acc);
        const float shift = ((const float *)l_bias)[EB_3 + 32 * nt + pl];
        float best = -__builtin_inff();
#pragma unroll
        for (int q = 0; q < TP; ++q) {
            if (a.feat != nullptr) {       // optional (B,512,N) output: 4 consecutive points per 16-byte store
                float *fo = a.feat + ((size_t)bi * EC4 + 32 * nt + pl) * N + tile0[q] + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p0 = tile0[q] + 4 * h + 8 * g;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[q][4 * g + e] + shift, 0.f);
                    if (p0 + 3 < N && (N & 3) == 0) {
                        *(f32x4 *)(fo + 8 * g) = f32x4{v[0], v[1], v[2], v[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (p0 + e < N) fo[8 * g + e] = v[e];
                    }
                }
            }
            if (tile0[q] + TILE > N) {     // ragged or empty tile: its missing points do not take part in the max
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile0[q] + (r & 3) + 8 * (r >> 2) + 4 * h >= N) acc[q][r] = -__builtin_inff();
            }
            float m = fmaxf(fmaxf(acc[q][0], acc[q][1]), acc[q][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[q][r]), acc[q][r + 1]);
            best = fmaxf(best, fmaxf(m, acc[q][15]));
        }
        best = half_max(best);
        // max_p relu(x_p + shift) = relu(max_p x_p + shift): fp32 addition is monotonic; an all-empty wave gives 0
        if (!h) l_wmax[wave * EC4 + 32 * nt + pl] = fmaxf(best + shift, 0.f);
    }
    __syncthreads();
    // ---- combine the waves, one integer atomicMax per feature (all values are >= 0, so the bit patterns order)
    for (int f = threadIdx.x; f < EC4; f += EW * 64) {
        float m = l_wmax[f];
#pragma unroll
        for (int w = 1; w < EW; ++w) m = fmaxf(m, l_wmax[w * EC4 + f]);
        atomicMax((int *)(a.gmax + (size_t)bi * EC4 + f), (int)f2u(m));
    }
}

int e_ns_of(int precision) {
    return precision == DPF_PREC_BF16 ? 1 : precision == DPF_PREC_BF16X3 ? 2 : precision == DPF_PREC_BF16X6 ? 3 : 0;
}

template <int NS>
int launch_enc(const EncArgs &a, hipStream_t s) {
    constexpr int EW = e_waves(NS), EWG_POINTS = EW * e_tp(NS) * TILE;
    const int lds = 8192 + 2 * ep_chunk_bytes(NS) + EW * EC4 * 4;
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)enc_kernel<NS>, lds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(enc_kernel<NS>, dim3((a.N + EWG_POINTS - 1) / EWG_POINTS, a.B), dim3(EW * 64), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t dpf_encoder_canon_floats(void) { return (size_t)E_CANON; }

extern "C" size_t dpf_encoder_packed_bytes(int precision) {
    const int ns = e_ns_of(precision);
    return ns ? ep_bytes(ns) : 0;
}

extern "C" int dpf_encoder_pack(int precision, const float *canon, void *packed, dpf_stream_t stream) {
    const int ns = e_ns_of(precision);
    if (!ns || !canon || !packed) return DPF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (ns == 1) hipLaunchKernelGGL(enc_pack_kernel<1>, dim3(256), dim3(256), 0, s, canon, (uint8_t *)packed);
    if (ns == 2) hipLaunchKernelGGL(enc_pack_kernel<2>, dim3(256), dim3(256), 0, s, canon, (uint8_t *)packed);
    if (ns == 3) hipLaunchKernelGGL(enc_pack_kernel<3>, dim3(256), dim3(256), 0, s, canon, (uint8_t *)packed);
    return (int)hipGetLastError();
}

extern "C" int dpf_encoder_forward(int B, int N, int precision, const void *packed, const float *x, float *gmax, float *feat,
                                   dpf_stream_t stream) {
    const int ns = e_ns_of(precision);
    if (!ns || B < 0 || N <= 0) return DPF_EINVAL;
    if (B == 0) return 0;
    if (!packed || !x || !gmax) return DPF_EINVAL;
    if (B > 65535) return DPF_ENOSUP;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = dpf_zero_async(gmax, sizeof(float) * (size_t)B * EC4, s);
    if (e != hipSuccess) return (int)e;
    EncArgs a{(const uint8_t *)packed, x, gmax, feat, B, N};
    return ns == 1 ? launch_enc<1>(a, s) : ns == 2 ? launch_enc<2>(a, s) : launch_enc<3>(a, s);
}
