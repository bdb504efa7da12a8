// The fused L-layer coupling stack on 16-POINT tiles, for small per-GPU batches (gfx950, MI355X).
//
// Same contract as flow_kernel in flow.hip (lib/networks/decoders.py:54-72, flows.py:95-160, layers.py:40-45, eval-mode
// BatchNorm) and the same packed canonical weights, FiLM blocks, precision (f16x3) and transform; what changes is the
// tiling.  flow_kernel gives a wave 32 points (v_mfma_f32_32x32x16): B = 32 clouds of 2048 points are 2048 tiles = two per
// SIMD, which is what that kernel is scheduled for.  A rank of an 8-GPU job that shards BASELINE's 32 clouds holds FOUR
// (SURVEY 8e): 256 tiles of 32 points leave three quarters of the SIMDs without a wave, and the launch takes as long as
// one wave needs for 14 layers whatever the batch (r04 measurement: 28.4 us at B = 4 and B = 8, 29.8 at B = 16).
// Here a wave owns 16 points and runs v_mfma_f32_16x16x32_{f16,bf16}: half the matrix time and half the per-lane VALU
// work per layer, twice the waves.
//
//   * weights = A operand (16 output features x 32 K), points = B operand (32 K x 16 points); the accumulator fragment --
//     lane (point n = lane & 15, group g = lane >> 4) holds features 16 t + 4 g + r, r = 0..3, of M tile t -- becomes, after
//     relu and the fp16 hi/lo split in registers, the B fragment of the next contraction: K slot (s, g, j) of k-step s is
//     feature 16 (2 s + (j >> 2)) + 4 g + (j & 3), a permutation applied to W1's columns at pack time
//     (pack16_kernel).  No LDS round trip for activations, no cross-lane traffic until the 4-group sum of the outputs
//     (v_permlane16_swap + v_permlane32_swap).
//   * a workgroup = CW compute waves (16 CW points of ONE cloud) + LW loader waves.  A layer's 42 KiB (32 KiB of W1
//     fragments, 8 KiB of input-layer fragments, the cloud's 2 KiB FiLM block) stream through a ring of three LDS slots
//     with global_load_lds; the loaders issue every piece (a piece stalls its issuer for 60-180 cycles, which a wave
//     that is alone on its SIMD cannot hide) and hand a slot over with ONE workgroup barrier per layer: after barrier n
//     every compute wave has finished layer n, so the slot of layer n is free for layer n + 3; before arriving at
//     barrier n the loaders wait (counted vmcnt) for the pieces of layer n + 1.
//   * per layer and wave: 8 input MFMAs, 48 chain MFMAs (3 fp16 products x 2 k-steps x 4 M tiles x 2 branches), 64 VALU of
//     relu + split, 64-96 of output contraction, the coupling transform.
#include <stdlib.h>

#include "flow_common.h"

#ifdef DPF_PROFILE
#define DPF16_T(i) { __builtin_amdgcn_sched_barrier(0); tt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define DPF16_T(i)
#endif

namespace {

__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// canonical fp32 layout (dpf_hip.h; the same constants as flow.hip)
constexpr int C16_W0 = 0, C16_BN0 = 128, C16_W1 = 384;
constexpr int C16_FILM = 4740;
__host__ __device__ constexpr int c16_film_floats(int G) { return 64 * G + 256 + 4096 + 64; }
__host__ __device__ constexpr int c16_branch_floats(int G) { return C16_FILM + 2 * c16_film_floats(G); }
__host__ __device__ constexpr int c16_layer_floats(int G) { return 2 * c16_branch_floats(G); }

// ---- pack: canonical weights -> 16-point-tile fragments (layout in flow_common.h) ----------------------------------
__global__ __launch_bounds__(256) void pack16_kernel(int G, const float *__restrict__ canon, uint8_t *__restrict__ packed16) {
    const int l = blockIdx.x;
    const float *cl = canon + (size_t)l * c16_layer_floats(G);
    uint16_t *o16 = (uint16_t *)(packed16 + (size_t)l * P16_LAYER);
    __shared__ float scratch[256];
    // the power of two of each branch's W1, as pack_kernel<2, true> (flow_common.h: w1_pow2_scale; the FiLM block carries its inverse)
    const float wsc0 = w1_pow2_scale(cl + 0 * c16_branch_floats(G) + C16_W1, scratch);
    const float wsc1 = w1_pow2_scale(cl + 1 * c16_branch_floats(G) + C16_W1, scratch);
    for (int idx = threadIdx.x; idx < 2 * 4 * 2 * 64 * 8; idx += blockDim.x) {       // [br][t'][s][lane][j]
        const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 1, tp = (idx >> 10) & 3, br = idx >> 12;
        const int i = lane & 15, g = lane >> 4;
        const int fi = 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3);
        const float w = cl[br * c16_branch_floats(G) + C16_W1 + (16 * tp + i) * 64 + fi] * (br ? wsc1 : wsc0);
        const _Float16 wh = (_Float16)w;                       // fp16 hi (RNE) + fp16 of the exact remainder, as pack_kernel<2, true>
        const _Float16 wl = (_Float16)(w - (float)wh);
        o16[idx] = __builtin_bit_cast(uint16_t, wh);
        o16[P16_A1_PART / 2 + idx] = __builtin_bit_cast(uint16_t, wl);
    }
    uint16_t *a0 = (uint16_t *)(packed16 + (size_t)l * P16_LAYER + P16_A1);
    for (int idx = threadIdx.x; idx < 2 * 4 * 64 * 8; idx += blockDim.x) {           // [br][t][lane][j]
        const int j = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) & 3, br = idx >> 11;
        const int f = 16 * t + (lane & 15), g = lane >> 4;
        uint32_t v = 0;
        if (g < 2) {
            const float *cb = cl + br * c16_branch_floats(G);
            const float gamma = cb[C16_BN0 + f], beta = cb[C16_BN0 + 64 + f], rm = cb[C16_BN0 + 128 + f], rv = cb[C16_BN0 + 192 + f];
            const float s0 = gamma / sqrtf(rv + BN_EPS);
            v = input_weight_slot(s0 * cb[C16_W0 + f * 2 + g], beta - rm * s0, g, j);
        }
        a0[idx] = (uint16_t)v;
    }
}

constexpr int L16_BYTES = P16_LAYER + FILM_BYTES;      // one ring slot: 43008 B
constexpr int NPIECE = P16_LAYER / 1024;               // 40 weight pieces of 1 KiB; + 2 FiLM pieces

template <int K, int PER>
__device__ __forceinline__ void issue16(const uint8_t *src, uint8_t *dst) {
    if constexpr (K < PER) {
        __builtin_amdgcn_global_load_lds((glb_void *)(src + (K / 4) * 4096), (lds_void *)(dst + (K / 4) * 4096), 16, (K % 4) * 1024, 0);
        issue16<K + 1, PER>(src, dst);
    }
}

// loader wave `lw` of LW: its contiguous run of the layer's weight pieces, and (loaders 0 and 1 % LW) a FiLM piece
template <int LW>
__device__ __forceinline__ void stage16(const FlowArgs &a, int li, int bi, uint8_t *slot, int lw, int lane) {
    constexpr int PER = NPIECE / LW;
    static_assert(NPIECE % LW == 0, "every loader takes the same run of pieces");
    const uint8_t *src = a.packed16 + (size_t)li * P16_LAYER + lw * (PER * 1024) + lane * 16;
    issue16<0, PER>(src, slot + lw * (PER * 1024));
#pragma unroll
    for (int f = 0; f < FILM_BYTES / 1024; ++f)
        if (lw == f % LW) {
            const uint8_t *fsrc = (const uint8_t *)a.film + ((size_t)li * a.B + bi) * FILM_BYTES + f * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((glb_void *)fsrc, (lds_void *)(slot + P16_LAYER + f * 1024), 16, 0, 0);
        }
}
// FiLM pieces loader `lw` issues per layer on top of its NPIECE / LW weight pieces
template <int LW>
__device__ __forceinline__ int film_pieces_of(int lw) {
    int e = 0;
#pragma unroll
    for (int f = 0; f < FILM_BYTES / 1024; ++f) e += (lw == f % LW) ? 1 : 0;
    return e;
}
// wait until only the pieces of ONE layer (this loader's BASE + extra) are outstanding: everything older has landed
template <int BASE>
__device__ __forceinline__ void wait_one_layer_left(int extra) {
    static_assert(BASE + 2 <= 63, "vmcnt is a 6-bit counter");
    if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");
    else if (extra == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + 2) : "memory");
}

// Sums over the four 16-lane groups that share a point, for the layer's four outputs at once.  v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of its second, so ONE swap + ONE add halves TWO values
// (the results land in different rows); v_permlane32_swap does the same with the wave's halves.  Three swaps + three adds
// leave row r with the total of value r (oa, ob, ma, mb); three more swaps hand every row all four.  Every lane adds the
// same operands in the same order: the four groups of a point agree bit for bit.
__device__ __forceinline__ void quad_sum4(float (&o)[2][2]) {
    const auto s1 = __builtin_amdgcn_permlane16_swap(f2u(o[0][0]), f2u(o[0][1]), false, false);
    const auto s2 = __builtin_amdgcn_permlane16_swap(f2u(o[1][0]), f2u(o[1][1]), false, false);
    const float x = u2f(s1[0]) + u2f(s1[1]);          // rows: oa(0+1) ob(0+1) oa(2+3) ob(2+3)
    const float y = u2f(s2[0]) + u2f(s2[1]);          //       ma(0+1) mb(0+1) ma(2+3) mb(2+3)
    const auto s3 = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(y), false, false);
    const float z = u2f(s3[0]) + u2f(s3[1]);          // rows: oa ob ma mb
    const auto t1 = __builtin_amdgcn_permlane16_swap(f2u(z), f2u(z), false, false);       // [oa oa ma ma], [ob ob mb mb]
    const auto ta = __builtin_amdgcn_permlane32_swap(t1[0], t1[0], false, false);         // [oa x4], [ma x4]
    const auto tb = __builtin_amdgcn_permlane32_swap(t1[1], t1[1], false, false);         // [ob x4], [mb x4]
    o[0][0] = u2f(ta[0]); o[1][0] = u2f(ta[1]);
    o[0][1] = u2f(tb[0]); o[1][1] = u2f(tb[1]);
}

// What a layer's head needs from its LDS slot, fetched while the PREVIOUS layer's transform runs (the slot is handed over by
// the barrier in the middle of that layer's tail): the input-layer fragments, branch A's first W1 fragments and shift.
struct Head16 {
    u32x4 a0[8];      // [br * 4 + t]
    u32x4 af0[8];     // branch A, k-step 0: [part * 4 + t']
    f32x4 d0[4];      // branch A's folded FiLM shift per M tile
};
// LDS addressing: `wl` = slot + 16 * lane (fragment reads), `fl` = slot + P16_LAYER + 16 * g (FiLM block reads), both
// laundered through an empty asm per layer so that every read is ONE register + a 16-bit immediate (left to itself the
// compiler materialises a separate loop-invariant address register for every (slot, offset) pair: ~100 VGPRs).
__device__ __forceinline__ u32x4 lds128(const uint8_t *smem, uint32_t base, int off) { return *(const u32x4 *)(smem + base + off); }
__device__ __forceinline__ f32x4 ldsf4(const uint8_t *smem, uint32_t base, int off) { return *(const f32x4 *)(smem + base + off); }

__device__ __forceinline__ void fetch_head(const uint8_t *smem, uint32_t wl, uint32_t fl, Head16 &h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) h.a0[i] = lds128(smem, wl, P16_A1 + i * 1024);
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) h.af0[part * 4 + tp] = lds128(smem, wl, part * P16_A1_PART + (tp * 2) * 1024);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) h.d0[tp] = ldsf4(smem, fl, 64 * tp);
}

// Both conditioner branches of one layer for one 16-point tile, up to the point where the layer's LDS slot is no longer
// needed: everything but branch B's output contraction, whose operands leave in registers (acc1b, cwa, cwb, sums).
//   G0  8 input MFMAs | A's two splits
//   G1  chain A, k-step 0 (12 MFMAs) | B's first split, fragments of the next k-step
//   G2  chain A, k-step 1            | B's second split
//   G3  chain B, k-step 0            | half of A's output contraction
//   G4  chain B, k-step 1            | the other half
struct Tail16 {
    f32x4 acc1b[4];   // branch B's pre-activations h1 + D per M tile
    f32x4 cwa[4], cwb[4];
    float pa[2][2], pb[2][2];
};
template <bool TWO>
__device__ __forceinline__ void layer16_main(const uint8_t *smem, uint32_t wl, uint32_t fl, const Head16 &hd, u32x4 b0, float negone, Tail16 &tl,
                                             unsigned long long *tt) {
    typedef Terms<2> TT;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0[2][4], acc1[2][4];
    u32x4 bfrag[2][2][2];                                          // [br][part][s]
    u32x4 af[2][8];                                                // [buffer][part * 4 + t']

    auto ld_frag = [&](int br, int s, u32x4 (&dst)[8]) {           // hi and lo fragments of the four M tiles of k-step s
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
                dst[part * 4 + tp] = lds128(smem, wl, part * P16_A1_PART + ((br * 4 + tp) * 2 + s) * 1024);
    };
    auto split_s = [&](int br, int s) {                            // relu + fp16 hi/lo split -> the B fragments of k-step s
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int t = 2 * s + (d >> 1), r = 2 * (d & 1);
            uint32_t hi_, lo_;
            split_relu_f16(acc0[br][t][r], acc0[br][t][r + 1], negone, hi_, lo_);
            bfrag[br][0][s][d] = hi_;
            bfrag[br][1][s][d] = lo_;
        }
    };
    auto chain_s = [&](int br, int s, const u32x4 (&a)[8]) {
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
                acc1[br][tp] = mfma16_f16(a[TT::A[term] * 4 + tp], bfrag[br][TT::B[term]][s], acc1[br][tp]);
    };
    {
        const f32x4 b2 = *(const f32x4 *)(smem + (fl & ~63u) + FILM_B2_OFF * 4);     // (fl - 16 g: the block's base)
        tl.pa[0][0] = 0.25f * b2.x; tl.pa[0][1] = 0.f;             // the output bias, a quarter in each lane group (exact)
        tl.pb[0][0] = TWO ? 0.25f * b2.y : 0.f; tl.pb[0][1] = 0.f;
        tl.pa[1][0] = 0.25f * b2.z; tl.pa[1][1] = 0.f;
        tl.pb[1][0] = TWO ? 0.25f * b2.w : 0.f; tl.pb[1][1] = 0.f;
    }
    auto contract_a = [&](int tp) {                                // o += W2' relu(h1 + D) over the 4 features of M tile tp, branch A
        const f32x4 wa4 = ldsf4(smem, fl, 256 + 64 * tp);
        const float r0 = relu(acc1[0][tp][0]), r1 = relu(acc1[0][tp][1]), r2 = relu(acc1[0][tp][2]), r3 = relu(acc1[0][tp][3]);
        tl.pa[0][0] = __builtin_fmaf(wa4.x, r0, tl.pa[0][0]); tl.pa[0][1] = __builtin_fmaf(wa4.y, r1, tl.pa[0][1]);
        tl.pa[0][0] = __builtin_fmaf(wa4.z, r2, tl.pa[0][0]); tl.pa[0][1] = __builtin_fmaf(wa4.w, r3, tl.pa[0][1]);
        if (TWO) {
            const f32x4 wb4 = ldsf4(smem, fl, 512 + 64 * tp);
            tl.pb[0][0] = __builtin_fmaf(wb4.x, r0, tl.pb[0][0]); tl.pb[0][1] = __builtin_fmaf(wb4.y, r1, tl.pb[0][1]);
            tl.pb[0][0] = __builtin_fmaf(wb4.z, r2, tl.pb[0][0]); tl.pb[0][1] = __builtin_fmaf(wb4.w, r3, tl.pb[0][1]);
        }
    };
    // `n` MFMAs, each followed by (the first `ds` of them) one LDS read and by `v` VALU
#define DPF16_PATTERN(n, v, ds)                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        if (i_ < (ds)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
        __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);                        \
    }
    // ---- G0: h0 pre-activations on the matrix core (fp32-accurate through the 3-way bf16 split), A's splits
#pragma unroll
    for (int br = 0; br < 2; ++br)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0[br][t] = mfma16_bf16(hd.a0[br * 4 + t], b0, z4);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) acc1[0][tp] = hd.d0[tp];
    ld_frag(0, 1, af[1]);
    split_s(0, 0);
    split_s(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    DPF16_T(1)
    // ---- G1, G2: chain A; B's splits ride behind its MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 0) ld_frag(1, 0, af[0]);                          // (af[0] = the head's fragments were copied out below)
        else {
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) acc1[1][tp] = ldsf4(smem, fl, FILM_BR_FLOATS * 4 + 64 * tp);
        }
        if (s == 0) chain_s(0, 0, hd.af0); else chain_s(0, 1, af[1]);
        split_s(1, s);
        DPF16_PATTERN(12, 2, 8)
        __builtin_amdgcn_sched_barrier(0);
    }
    DPF16_T(2)
    // ---- G3, G4: chain B; A's output contraction rides behind its MFMAs; B's contraction weights leave in registers
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 0) ld_frag(1, 1, af[1]);
        else {
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                tl.cwa[tp] = ldsf4(smem, fl, FILM_BR_FLOATS * 4 + 256 + 64 * tp);
                if (TWO) tl.cwb[tp] = ldsf4(smem, fl, FILM_BR_FLOATS * 4 + 512 + 64 * tp);
            }
        }
        chain_s(1, s, af[s]);
        contract_a(2 * s);
        contract_a(2 * s + 1);
        DPF16_PATTERN(12, (TWO ? 3 : 2), 8)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef DPF16_PATTERN
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) tl.acc1b[tp] = acc1[1][tp];
}

// branch B's output contraction from registers, then the four pre-activation outputs per lane group
template <bool TWO>
__device__ __forceinline__ void layer16_tail(Tail16 &tl, float (&o)[2][2]) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const f32x4 wa4 = tl.cwa[tp];
        const float r0 = relu(tl.acc1b[tp][0]), r1 = relu(tl.acc1b[tp][1]), r2 = relu(tl.acc1b[tp][2]), r3 = relu(tl.acc1b[tp][3]);
        tl.pa[1][0] = __builtin_fmaf(wa4.x, r0, tl.pa[1][0]); tl.pa[1][1] = __builtin_fmaf(wa4.y, r1, tl.pa[1][1]);
        tl.pa[1][0] = __builtin_fmaf(wa4.z, r2, tl.pa[1][0]); tl.pa[1][1] = __builtin_fmaf(wa4.w, r3, tl.pa[1][1]);
        if (TWO) {
            const f32x4 wb4 = tl.cwb[tp];
            tl.pb[1][0] = __builtin_fmaf(wb4.x, r0, tl.pb[1][0]); tl.pb[1][1] = __builtin_fmaf(wb4.y, r1, tl.pb[1][1]);
            tl.pb[1][0] = __builtin_fmaf(wb4.z, r2, tl.pb[1][0]); tl.pb[1][1] = __builtin_fmaf(wb4.w, r3, tl.pb[1][1]);
        }
    }
#pragma unroll
    for (int br = 0; br < 2; ++br) { o[br][0] = tl.pa[br][0] + tl.pa[br][1]; o[br][1] = tl.pb[br][0] + tl.pb[br][1]; }
}

template <int CW, int LW, bool INV>
__global__ __launch_bounds__((CW + LW) * 64) void flow16_kernel(FlowArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, g = lane >> 4, pl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, L = a.L;
    constexpr bool inverse = INV;

    if (wave >= CW) {
        // ================= loader wave: the ring's producer =================
        const int lw = wave - CW;
        const int extra = film_pieces_of<LW>(lw);
        auto issue = [&](int st) {
            if (st < L) stage16<LW>(a, inverse ? L - 1 - st : st, bi, smem + (st % 3) * L16_BYTES, lw, lane);
        };
        issue(0);
        issue(1);
        if (L > 1) wait_one_layer_left<NPIECE / LW>(extra);      // layer 0 has landed: only layer 1's pieces may be outstanding
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // barrier P
        for (int step = 0; step < L; ++step) {
            // barrier `step - 1` is behind us: layer step - 1's slot is free -> layer step + 2
            issue(step + 2);
            // layer step + 1 must have landed before barrier `step` releases the compute waves into it
            if (step + 2 < L) wait_one_layer_left<NPIECE / LW>(extra);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // barrier `step`
        }
        return;
    }

    // ================= compute wave: 16 points of cloud bi through all L layers =================
    const int n = (blockIdx.x * CW + wave) * 16 + pl;
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
        if (a.z_out != nullptr && valid && g == 0) {
            a.z_out[cloud + n] = p0; a.z_out[cloud + N + n] = p1; a.z_out[cloud + 2 * (size_t)N + n] = p2;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;   // running sum of logvar per channel
    const size_t list_stride = (size_t)a.B * 3 * N;
    float negone = -1.0f;                      // opaque to the compiler: see split_relu_f16
    asm volatile("" : "+s"(negone));
    float zero_lv = 0.0f;
    asm volatile("" : "+s"(zero_lv));
    const float v_keep = a.eps + __expf(zero_lv);
    const float k_keep = inverse ? __builtin_amdgcn_rsqf(v_keep) : __builtin_amdgcn_sqrtf(v_keep);
    const int lfirst = inverse ? L - 1 : 0;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // barrier P: layer 0 is in slot 0
    Head16 hd;
    auto slot_bases = [&](int st, uint32_t &wl, uint32_t &fl) {
        const uint32_t so = (uint32_t)(st % 3) * L16_BYTES;
        wl = so + lane * 16;
        fl = so + P16_LAYER + g * 16;
        asm volatile("" : "+v"(wl), "+v"(fl));
    };
    {
        uint32_t wl, fl;
        slot_bases(0, wl, fl);
        fetch_head(smem, wl, fl, hd);
    }

    for (int step = 0; step < L; ++step) {
        const int li = inverse ? L - 1 - step : step;
        const int ln = inverse ? (li > 0 ? li - 1 : 0) : (li + 1 < L ? li + 1 : li);
        uint32_t wl, fl;
        slot_bases(step, wl, fl);
        int nka, nkb, nwa, nwb;
        layer_meta(ln, nka, nkb, nwa, nwb);
        // ---- B operand of the input MFMA: group 0 carries kept channel a, group 1 kept channel b, groups 2 and 3 nothing
        const float xa = sel3(ka, p0, p1, p2);
        const float xb = kb < 0 ? 0.f : sel3(kb, p0, p1, p2);
        u32x4 b0 = input_fragment(g == 1 ? xb : xa, g == 1 ? 1 : 0);
        if (g >= 2) { b0.x = 0u; b0.y = 0u; b0.z = 0u; b0.w = 0u; }

        Tail16 tl;
        unsigned long long tt[8];
        (void)tt;
        DPF16_T(0)
        if (wb < 0) layer16_main<false>(smem, wl, fl, hd, b0, negone, tl, tt);
        else layer16_main<true>(smem, wl, fl, hd, b0, negone, tl, tt);
        DPF16_T(3)
        // barrier `step`: every LDS read of this layer's slot has returned (its values are in registers), so the slot is
        // free for layer step + 3; the loaders have waited for layer step + 1, whose head is fetched under the tail below
        __syncthreads();
        DPF16_T(4)
        // (unconditional: a conditional fetch would keep the OLD head's 80 registers alive through the whole layer as the
        // other arm of a select; after the last layer this reads a slot nobody needs and the values are dropped)
        slot_bases(step + 1, wl, fl);
        fetch_head(smem, wl, fl, hd);
        float o[2][2];
        if (wb < 0) layer16_tail<false>(tl, o); else layer16_tail<true>(tl, o);
        quad_sum4(o);
        DPF16_T(5)
        // ---- coupling transform (flows.py:96-115), exactly flow_kernel's
        float lva, lvb = 0.f, fa, fb = k_keep;
        lva = o[0][0] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][0]));       // softsign, :99
        const float va = a.eps + __expf(lva);
        fa = inverse ? __builtin_amdgcn_rsqf(va) : __builtin_amdgcn_sqrtf(va);
        if (wb >= 0) {
            lvb = o[0][1] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][1]));
            const float vb = a.eps + __expf(lvb);
            fb = inverse ? __builtin_amdgcn_rsqf(vb) : __builtin_amdgcn_sqrtf(vb);
        }
        float lv[3], mu[3], pn[3];
        const float pin[3] = {p0, p1, p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lv[c] = c == wa ? lva : (c == wb ? lvb : 0.f);
            mu[c] = c == wa ? o[1][0] : (c == wb ? o[1][1] : 0.f);
            const float f = c == wa ? fa : (c == wb ? fb : k_keep);
            pn[c] = inverse ? (pin[c] - mu[c]) * f : f * pin[c] + mu[c];
        }
        p0 = pn[0]; p1 = pn[1]; p2 = pn[2];
        s0 += lv[0]; s1 += lv[1]; s2 += lv[2];
        if (a.ps != nullptr && valid && g < 3) {   // per-layer lists in DIRECT order: group 0 the points, 1 the means, 2 the log-variances
            const size_t base = (size_t)li * list_stride + cloud + n;
            float *dst = g == 0 ? a.ps : (g == 1 ? a.mus : a.lvs);
            const float v0 = g == 0 ? pn[0] : (g == 1 ? mu[0] : lv[0]);
            const float v1 = g == 0 ? pn[1] : (g == 1 ? mu[1] : lv[1]);
            const float v2 = g == 0 ? pn[2] : (g == 1 ? mu[2] : lv[2]);
            dst[base] = v0; dst[base + N] = v1; dst[base + 2 * (size_t)N] = v2;
        }
        ka = nka; kb = nkb; wa = nwa; wb = nwb;
#ifdef DPF_PROFILE
        DPF16_T(6)
        if (a.prof != nullptr && lane == 0 && blockIdx.x < 2 && blockIdx.y == 0) {
            unsigned long long *o2 = a.prof + (((size_t)(blockIdx.x * CW + wave)) * L + step) * 8;
            for (int i = 0; i < 7; ++i) o2[i] = tt[i];
        }
#endif
    }
    if (valid) {
        if (g == 0) {
            a.p_out[cloud + n] = p0; a.p_out[cloud + N + n] = p1; a.p_out[cloud + 2 * (size_t)N + n] = p2;
        } else if (g == 1 && a.p_out_pm != nullptr) {   // point-major (B,N,3) copy for the structural losses (evaluating.py:110)
            float *o2 = a.p_out_pm + ((size_t)bi * N + n) * 3;
            o2[0] = p0; o2[1] = p1; o2[2] = p2;
        } else if (g == 2 && a.sum_lv != nullptr) {
            a.sum_lv[cloud + n] = s0; a.sum_lv[cloud + N + n] = s1; a.sum_lv[cloud + 2 * (size_t)N + n] = s2;
        }
    }
}

// ======================================================================================================================
// Branch-split variant, for batches that leave most SIMDs EMPTY even with 16-point tiles (B * N / 16 <= 512 tiles: the four
// clouds a rank of an 8-GPU job holds of BASELINE's 32): the two conditioner branches of a tile -- they depend on the
// layer input only -- run in TWO waves on two SIMDs.  Each wave runs 4 input MFMAs, one relu + split, one 24-MFMA chain and
// one output contraction instead of two, reduces its two outputs over the four lane groups, leaves them in a small LDS
// exchange buffer in front of the layer's (only) workgroup barrier, picks up its partner's behind it, and both apply the
// coupling transform to their own copy of the points (same inputs, same instructions: the copies stay bit-identical).
// The exchange buffer alternates between two halves by layer parity: a wave may write layer n + 1's outputs while its
// partner has not yet read layer n's.
struct Head16S {
    u32x4 a0[4];      // this branch's input-layer fragments per M tile
    u32x4 af0[8];     // this branch's W1 fragments of k-step 0: [part * 4 + t']
    f32x4 d0[4];      // this branch's folded FiLM shift per M tile
};
__device__ __forceinline__ void fetch_head_s(const uint8_t *smem, uint32_t w1, uint32_t w0, uint32_t fb, Head16S &h) {
#pragma unroll
    for (int t = 0; t < 4; ++t) h.a0[t] = lds128(smem, w0, t * 1024);
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) h.af0[part * 4 + tp] = lds128(smem, w1, part * P16_A1_PART + (tp * 2) * 1024);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) h.d0[tp] = ldsf4(smem, fb, 64 * tp);
}

// one conditioner branch of one layer for one 16-point tile -> its two pre-activation outputs, summed over the lane groups
template <bool TWO>
__device__ __forceinline__ void branch16(const uint8_t *smem, uint32_t w1, uint32_t fb, float bias_a, float bias_b, const Head16S &hd,
                                         u32x4 b0, float negone, float &oa, float &ob) {
    typedef Terms<2> TT;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0[4], acc1[4], cwa[4], cwb[4];
    u32x4 bfrag[2][2], af1[8];
    auto split_s = [&](int s) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int t = 2 * s + (d >> 1), r = 2 * (d & 1);
            uint32_t hi_, lo_;
            split_relu_f16(acc0[t][r], acc0[t][r + 1], negone, hi_, lo_);
            bfrag[0][s][d] = hi_;
            bfrag[1][s][d] = lo_;
        }
    };
    auto chain_s = [&](int s, const u32x4 (&a)[8]) {
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) acc1[tp] = mfma16_f16(a[TT::A[term] * 4 + tp], bfrag[TT::B[term]][s], acc1[tp]);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) acc0[t] = mfma16_bf16(hd.a0[t], b0, z4);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) acc1[tp] = hd.d0[tp];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) af1[part * 4 + tp] = lds128(smem, w1, part * P16_A1_PART + (tp * 2 + 1) * 1024);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        cwa[tp] = ldsf4(smem, fb, 256 + 64 * tp);
        if (TWO) cwb[tp] = ldsf4(smem, fb, 512 + 64 * tp);
    }
    split_s(0);
    __builtin_amdgcn_sched_barrier(0);
    chain_s(0, hd.af0);
    split_s(1);
#pragma unroll
    for (int i_ = 0; i_ < 12; ++i_) {                              // k-step 1's split rides behind k-step 0's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    chain_s(1, af1);
    float pa0 = bias_a, pa1 = 0.f, pb0 = bias_b, pb1 = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const f32x4 wa4 = cwa[tp];
        const float r0 = relu(acc1[tp][0]), r1 = relu(acc1[tp][1]), r2 = relu(acc1[tp][2]), r3 = relu(acc1[tp][3]);
        pa0 = __builtin_fmaf(wa4.x, r0, pa0); pa1 = __builtin_fmaf(wa4.y, r1, pa1);
        pa0 = __builtin_fmaf(wa4.z, r2, pa0); pa1 = __builtin_fmaf(wa4.w, r3, pa1);
        if (TWO) {
            const f32x4 wb4 = cwb[tp];
            pb0 = __builtin_fmaf(wb4.x, r0, pb0); pb1 = __builtin_fmaf(wb4.y, r1, pb1);
            pb0 = __builtin_fmaf(wb4.z, r2, pb0); pb1 = __builtin_fmaf(wb4.w, r3, pb1);
        }
    }
    const float o0 = pa0 + pa1, o1 = pb0 + pb1;
    // the two sums over the four lane groups: one swap + add per stage for both values, then hand every group both
    const auto s1 = __builtin_amdgcn_permlane16_swap(f2u(o0), f2u(o1), false, false);
    const float x = u2f(s1[0]) + u2f(s1[1]);          // rows: o0(0+1) o1(0+1) o0(2+3) o1(2+3)
    const auto s2 = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    const float z = u2f(s2[0]) + u2f(s2[1]);          // rows: o0 o1 o0 o1
    const auto t1 = __builtin_amdgcn_permlane16_swap(f2u(z), f2u(z), false, false);
    oa = u2f(t1[0]);
    ob = u2f(t1[1]);
}

template <int CW, int LW, bool INV>
__global__ __launch_bounds__((CW + LW) * 64) void flow16s_kernel(FlowArgs a) {
    static_assert(CW % 2 == 0, "two waves per tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int T = CW / 2;                                      // tiles per workgroup
    float *xbuf = (float *)(smem + 3 * L16_BYTES);                 // [parity][tile][br][value][16 points]
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, g = lane >> 4, pl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, L = a.L;
    constexpr bool inverse = INV;

    if (wave >= CW) {          // loader wave: exactly flow16_kernel's
        const int lw = wave - CW;
        const int extra = film_pieces_of<LW>(lw);
        auto issue = [&](int st) {
            if (st < L) stage16<LW>(a, inverse ? L - 1 - st : st, bi, smem + (st % 3) * L16_BYTES, lw, lane);
        };
        issue(0);
        issue(1);
        if (L > 1) wait_one_layer_left<NPIECE / LW>(extra);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int step = 0; step < L; ++step) {
            issue(step + 2);
            if (step + 2 < L) wait_one_layer_left<NPIECE / LW>(extra);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }

    const int tile = wave >> 1, br = wave & 1;                     // br 0 = logvar branch, 1 = mu branch
    const int n = (blockIdx.x * T + tile) * 16 + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const size_t cloud = (size_t)bi * 3 * N;
    float p0 = a.p_in[cloud + nc], p1 = a.p_in[cloud + N + nc], p2 = a.p_in[cloud + 2 * (size_t)N + nc];
    if (a.base_mu != nullptr) {
        float *pp[3] = {&p0, &p1, &p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float lv = a.base_lv[bi * a.lv_sb + c * a.lv_sc + nc * a.lv_sn];
            const float mu = a.base_mu[bi * a.mu_sb + c * a.mu_sc + nc * a.mu_sn];
            *pp[c] = __fadd_rn(__fmul_rn(*pp[c], expf(__fmul_rn(0.5f, lv))), mu);
        }
        if (a.z_out != nullptr && valid && g == 0 && br == 0) {
            a.z_out[cloud + n] = p0; a.z_out[cloud + N + n] = p1; a.z_out[cloud + 2 * (size_t)N + n] = p2;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const size_t list_stride = (size_t)a.B * 3 * N;
    float negone = -1.0f;
    asm volatile("" : "+s"(negone));
    float zero_lv = 0.0f;
    asm volatile("" : "+s"(zero_lv));
    const float v_keep = a.eps + __expf(zero_lv);
    const float k_keep = inverse ? __builtin_amdgcn_rsqf(v_keep) : __builtin_amdgcn_sqrtf(v_keep);
    const int lfirst = inverse ? L - 1 : 0;
    auto pack_meta = [&](int row) {
        const int4 m = ((const int4 *)a.meta)[min(row, L - 1)];
        return (m.x + 1) | ((m.y + 1) << 2) | ((m.z + 1) << 4) | ((m.w + 1) << 6);
    };
    const int code_lo = pack_meta(lane), code_hi = pack_meta(64 + lane);
    auto layer_meta = [&](int l, int &k0, int &k1, int &w0, int &w1) {
        const int c = __builtin_amdgcn_readlane(l >= 64 ? code_hi : code_lo, l & 63);
        k0 = (c & 3) - 1; k1 = ((c >> 2) & 3) - 1; w0 = ((c >> 4) & 3) - 1; w1 = ((c >> 6) & 3) - 1;
    };
    int ka, kb, wa, wb;
    layer_meta(lfirst, ka, kb, wa, wb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                               // barrier P
    // this branch's slices of a slot: W1 fragments, input-layer fragments, FiLM vectors (see lds128's note)
    auto slot_bases = [&](int st, uint32_t &w1, uint32_t &w0, uint32_t &fb) {
        const uint32_t so = (uint32_t)(st % 3) * L16_BYTES;
        w1 = so + br * 8192 + lane * 16;
        w0 = so + P16_A1 + br * 4096 + lane * 16;
        fb = so + P16_LAYER + br * (FILM_BR_FLOATS * 4) + g * 16;
        asm volatile("" : "+v"(w1), "+v"(w0), "+v"(fb));
    };
    Head16S hd;
    float bias_a, bias_b;
    auto fetch = [&](int st) {
        uint32_t w1, w0, fb;
        slot_bases(st, w1, w0, fb);
        fetch_head_s(smem, w1, w0, fb, hd);
        const float *b2 = (const float *)(smem + (uint32_t)(st % 3) * L16_BYTES + P16_LAYER) + FILM_B2_OFF + br * 2;
        bias_a = 0.25f * b2[0];                                    // the output bias, a quarter in each lane group (exact)
        bias_b = 0.25f * b2[1];
    };
    fetch(0);

    for (int step = 0; step < L; ++step) {
        const int li = inverse ? L - 1 - step : step;
        const int ln = inverse ? (li > 0 ? li - 1 : 0) : (li + 1 < L ? li + 1 : li);
        uint32_t w1, w0, fb;
        slot_bases(step, w1, w0, fb);
        unsigned long long tt[8];
        (void)tt;
        DPF16_T(0)
        int nka, nkb, nwa, nwb;
        layer_meta(ln, nka, nkb, nwa, nwb);
        const float xa = sel3(ka, p0, p1, p2);
        const float xb = kb < 0 ? 0.f : sel3(kb, p0, p1, p2);
        u32x4 b0 = input_fragment(g == 1 ? xb : xa, g == 1 ? 1 : 0);
        if (g >= 2) { b0.x = 0u; b0.y = 0u; b0.z = 0u; b0.w = 0u; }
        DPF16_T(1)

        float mine0, mine1;
        if (wb < 0) branch16<false>(smem, w1, fb, bias_a, 0.f, hd, b0, negone, mine0, mine1);
        else branch16<true>(smem, w1, fb, bias_a, bias_b, hd, b0, negone, mine0, mine1);
        DPF16_T(2)
        float *xw = xbuf + (((step & 1) * T + tile) * 2 + br) * 32;
        if (g == 0) { xw[pl] = mine0; xw[16 + pl] = mine1; }
        __syncthreads();       // barrier `step`: outputs exchanged; this layer's slot is free; the next layer has landed
        DPF16_T(3)
        fetch(step + 1);       // (unconditional, as in flow16_kernel)
        const float *xr = xbuf + (((step & 1) * T + tile) * 2 + (br ^ 1)) * 32;
        const float their0 = xr[pl], their1 = xr[16 + pl];
        float o[2][2];
        o[0][0] = br ? their0 : mine0; o[0][1] = br ? their1 : mine1;
        o[1][0] = br ? mine0 : their0; o[1][1] = br ? mine1 : their1;
        // ---- coupling transform (flows.py:96-115), exactly flow_kernel's
        float lva, lvb = 0.f, fa, fb2 = k_keep;
        lva = o[0][0] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][0]));
        const float va = a.eps + __expf(lva);
        fa = inverse ? __builtin_amdgcn_rsqf(va) : __builtin_amdgcn_sqrtf(va);
        if (wb >= 0) {
            lvb = o[0][1] * __builtin_amdgcn_rcpf(1.0f + fabsf(o[0][1]));
            const float vb = a.eps + __expf(lvb);
            fb2 = inverse ? __builtin_amdgcn_rsqf(vb) : __builtin_amdgcn_sqrtf(vb);
        }
        float lv[3], mu[3], pn[3];
        const float pin[3] = {p0, p1, p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lv[c] = c == wa ? lva : (c == wb ? lvb : 0.f);
            mu[c] = c == wa ? o[1][0] : (c == wb ? o[1][1] : 0.f);
            const float f = c == wa ? fa : (c == wb ? fb2 : k_keep);
            pn[c] = inverse ? (pin[c] - mu[c]) * f : f * pin[c] + mu[c];
        }
        p0 = pn[0]; p1 = pn[1]; p2 = pn[2];
        s0 += lv[0]; s1 += lv[1]; s2 += lv[2];
        if (a.ps != nullptr && valid && g < 2) {   // lists: the logvar wave stores points (group 0) and log-variances (1), the mu wave the means (0)
            const size_t base = (size_t)li * list_stride + cloud + n;
            float *dst = br ? a.mus : (g == 0 ? a.ps : a.lvs);
            const float v0 = br ? mu[0] : (g == 0 ? pn[0] : lv[0]);
            const float v1 = br ? mu[1] : (g == 0 ? pn[1] : lv[1]);
            const float v2 = br ? mu[2] : (g == 0 ? pn[2] : lv[2]);
            if (!(br && g == 1)) { dst[base] = v0; dst[base + N] = v1; dst[base + 2 * (size_t)N] = v2; }
        }
        DPF16_T(4)
#ifdef DPF_PROFILE
        if (a.prof != nullptr && lane == 0 && blockIdx.x < 2 && blockIdx.y == 0) {
            unsigned long long *o2 = a.prof + (((size_t)(blockIdx.x * CW + wave)) * L + step) * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) o2[i] = tt[i];
        }
#endif
        ka = nka; kb = nkb; wa = nwa; wb = nwb;
    }
    if (valid) {
        if (br == 0 && g == 0) {
            a.p_out[cloud + n] = p0; a.p_out[cloud + N + n] = p1; a.p_out[cloud + 2 * (size_t)N + n] = p2;
        } else if (br == 1 && g == 0 && a.p_out_pm != nullptr) {
            float *o2 = a.p_out_pm + ((size_t)bi * N + n) * 3;
            o2[0] = p0; o2[1] = p1; o2[2] = p2;
        } else if (br == 0 && g == 1 && a.sum_lv != nullptr) {
            a.sum_lv[cloud + n] = s0; a.sum_lv[cloud + N + n] = s1; a.sum_lv[cloud + 2 * (size_t)N + n] = s2;
        }
    }
}

template <int CW, int LW, bool INV>
int launch16s(const FlowArgs &a, hipStream_t s) {
    const int lds = 3 * L16_BYTES + 2 * (CW / 2) * 2 * 32 * (int)sizeof(float);
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)flow16s_kernel<CW, LW, INV>, lds); e != hipSuccess) return (int)e;
    const dim3 grid((a.N + 16 * (CW / 2) - 1) / (16 * (CW / 2)), a.B), block((CW + LW) * 64);
    hipLaunchKernelGGL((flow16s_kernel<CW, LW, INV>), grid, block, lds, s, a);
    return (int)hipGetLastError();
}

template <int CW, int LW, bool INV>
int launch16(const FlowArgs &a, hipStream_t s) {
    const int lds = 3 * L16_BYTES;
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)flow16_kernel<CW, LW, INV>, lds); e != hipSuccess) return (int)e;
    const dim3 grid((a.N + 16 * CW - 1) / (16 * CW), a.B), block((CW + LW) * 64);
    hipLaunchKernelGGL((flow16_kernel<CW, LW, INV>), grid, block, lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// 16-point tiles pay while they leave SIMDs a single wave: up to 1024 tiles (one per SIMD of the 256 CUs); f16x3 only; the
// training forward's moment epilogue lives in the 32-point kernel.  DPF_FLOW_TILE16=0 / 1 forces the choice.
static int g_tile16_mode = getenv("DPF_FLOW_TILE16") ? atoi(getenv("DPF_FLOW_TILE16")) : -1;
// -1 (default): by size; 0: never; 1: whenever the precision allows (tests, tools/flow_sweep.py).  Returns the old mode.
extern "C" int dpf_flow_set_tile16(int mode) {
    const int old = g_tile16_mode;
    g_tile16_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return old;
}
bool flow16_serves(int n_layers, int B, int N, int precision, bool has_xs) {
    const int env = g_tile16_mode;
    if (precision != DPF_PREC_F16X3 || has_xs || env == 0 || n_layers > 128 || B > 65535) return false;
    if (env == 1) return true;
    return (long)B * ((N + 15) / 16) <= 1024;
}

static long g_tile16_launches = 0;
extern "C" long dpf_flow_tile16_launches(void) { return g_tile16_launches; }   // how many calls the 16-point kernel served

int flow16_launch(const void *flow_args, hipStream_t stream) {
    const FlowArgs &a = *(const FlowArgs *)flow_args;
    ++g_tile16_launches;
    static const int cw_env = getenv("DPF_FLOW16_CW") ? atoi(getenv("DPF_FLOW16_CW")) : 0;
    // compute waves per workgroup (the loaders share their SIMDs): 4 = one per SIMD; 2 when that is what gives every CU a
    // workgroup (8 + 4 waves would have to live in 168 registers each: the kernel spills there, so it is not built)
    int cw = cw_env ? cw_env : 4;
    if (!cw_env && (long)a.B * ((a.N + 63) / 64) < 160) cw = 2;
    const bool inv = a.mode == DPF_MODE_INVERSE;
    // at most half a tile per SIMD: split every tile's two branches over two waves (DPF_FLOW16_SPLIT=0 / 1 forces)
    static const int split_env = getenv("DPF_FLOW16_SPLIT") ? atoi(getenv("DPF_FLOW16_SPLIT")) : -1;
    const bool split = split_env >= 0 ? split_env != 0 : (long)a.B * ((a.N + 15) / 16) <= 512;
    if (split) {
        return inv ? launch16s<4, 4, true>(a, stream) : launch16s<4, 4, false>(a, stream);
    }
    if (cw >= 4) return inv ? launch16<4, 4, true>(a, stream) : launch16<4, 4, false>(a, stream);
    return inv ? launch16<2, 2, true>(a, stream) : launch16<2, 2, false>(a, stream);
}

// called by dpf_flow_pack (flow.hip) for the f16x3 precision
int flow16_pack(int n_layers, int G, const float *canon, void *packed16, hipStream_t s) {
    hipLaunchKernelGGL(pack16_kernel, dim3(n_layers), dim3(256), 0, s, G, canon, (uint8_t *)packed16);
    return (int)hipGetLastError();
}
