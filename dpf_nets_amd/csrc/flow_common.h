// Shared device helpers of the flow kernels (csrc/flow.hip: eval-mode fused stack;
// csrc/flow_train.hip: training-mode per-layer forward/backward).  Everything is in an
// anonymous namespace: each translation unit gets its own copy.
#ifndef DPF_FLOW_COMMON_H
#define DPF_FLOW_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpf_hip.h"
#include "lds_attr.h"

namespace {

static_assert(DPF_FLOW_F == 64, "kernels are built for 64 hidden features");
constexpr float BN_EPS = 1e-5f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- packed layout ----------------------------------------------------------
constexpr int P_A1_PART = 16384;                       // [br2][t2][s4][lane64][8 bf16]
__host__ __device__ constexpr int p_a0_off(int NS) { return NS * P_A1_PART; }            // [br2][t2][lane64][8 bf16]
__host__ __device__ constexpr int p_layer_bytes(int NS) { return NS * P_A1_PART + 4096; }
constexpr int FILM_BYTES = 2048;                       // per (layer, cloud): [br2]{D[64], Wa[64], Wb[64]}, b2[br2][2], pad
constexpr int FILM_BR_FLOATS = 192;
constexpr int FILM_B2_OFF = 384;                       // floats

__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t bf16_rne(float x) {   // top-16 bits, round to nearest even
    const uint32_t u = f2u(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// relu on the bit pattern: one v_max_i32 (fmaxf would add a canonicalising v_max per MFMA output)
__device__ __forceinline__ float relu(float x) { return u2f((uint32_t)max((int)f2u(x), 0)); }
// {bf16(a), bf16(b)} round-to-nearest-even in one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16_rne(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// {top16(a), top16(b)} (truncation) in one v_perm_b32
__device__ __forceinline__ uint32_t pack_bf16_trunc(float a, float b) {
    return __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u);
}
// truncation split: x = hi + rest exactly, hi has 8 significant bits
__device__ __forceinline__ uint32_t split_hi(float x, float &rest) {
    const uint32_t h = f2u(x) & 0xFFFF0000u;
    rest = x - u2f(h);
    return h;
}

constexpr int TILE = 32;          // points per tile (one MFMA N tile)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// fp16 hi/lo split of a pair with the ReLU folded in (operands of the f16x3 contraction; csrc/flow.hip layer_pipe and
// the training kernels' recomputation use this one function, so their fragments are bit-identical):
//   hi = relu(x) truncated to fp16 -- the RAW pair is truncated toward zero (one v_cvt_pkrtz_f16_f32) and one
//        v_pk_max_f16 zeroes the negative halves;
//   lo = fp16(x - hi): for x < 0 the remainder is <= 0 and the clamp modifier ([0, 1]) of v_fma_mixlo/hi_f16 makes it 0,
//        for x >= 0 it lies in [0, ulp(hi)) -- below 1 for |x| < 1024 -- and passes.
// `negone` must be -1.0 in an SGPR the compiler cannot see through (a literal -1 turns the fma into a subtraction of an
// extended half, which does not select the mix instructions).  4 VALU per pair.
__device__ __forceinline__ void split_relu_f16(float x0, float x1, float negone, uint32_t &hi, uint32_t &lo) {
    const fp16x2 hr = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const f16x2 lo_raw = {(_Float16)__builtin_fmaf((float)hr[0], negone, x0), (_Float16)__builtin_fmaf((float)hr[1], negone, x1)};
    const f16x2 one2 = {(_Float16)1.f, (_Float16)1.f}, zero2 = {(_Float16)0.f, (_Float16)0.f};
    lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(lo_raw, zero2), one2));
    hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, hr), zero2));
}
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float sel3(int c, float v0, float v1, float v2) {   // c wave-uniform
    return c == 0 ? v0 : (c == 1 ? v1 : v2);
}

// split-precision product terms: parts are 0 = hi, 1 = mid/lo, 2 = lo
template <int NS> struct Terms;
template <> struct Terms<1> { static constexpr int N = 1; static constexpr int A[1] = {0}, B[1] = {0}; };
template <> struct Terms<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0}; };
template <> struct Terms<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {1, 2, 0, 1, 0, 0}, B[6] = {1, 0, 2, 0, 1, 0};
};

// r05: f16x3 packs W1 as fp16 hi + fp16 lo.  At the reference's init scale (|W1| ~ 0.04) the lo parts are fp16
// SUBNORMALS (17-19 significant bits instead of 22), and a branch whose weights are small altogether loses the hi part too.
// W1 is therefore packed times a power of two per (layer, branch) that puts its largest entry into [2^13, 2^14): hi AND lo of
// every entry within 2^-11 of the largest are normal numbers.  The contraction then yields 2^k (h1 + D) -- the FiLM block
// carries D 2^k and W2' 2^-k (exact scalings, made where the block is made: film_kernel / tfold_kernel) -- so the eval
// kernels do not change, and the training kernels unscale where they take sums of the pre-activation.
// Block-wide: every thread of a 256-thread block calls it with the same W1 (4096 floats); returns the scale in all threads.
__device__ __forceinline__ float w1_pow2_scale(const float *__restrict__ W1, float *scratch /* 256 floats of LDS */) {
    float m = 0.f;
    for (int i = threadIdx.x; i < 4096; i += 256) m = fmaxf(m, fabsf(W1[i]));
    __syncthreads();                               // (scratch may still be read from a previous call)
    scratch[threadIdx.x] = m;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t = fmaxf(t, scratch[i]);        // every thread the same maximum
    const int e = (int)((f2u(t) >> 23) & 0xFFu) - 127;             // t in [2^e, 2^(e+1))
    const bool ok = t > 0.f && e > -100 && e < 100;                // (all zeros, denormal-tiny, Inf / NaN: no scaling)
    return ok ? u2f((uint32_t)(127 + 13 - e) << 23) : 1.0f;
}

// One conditioner branch (logvar or mu) of one layer for one 32-point tile:
// returns the two pre-activation outputs o_a, o_b of the branch (sum over this
// lane-half's 32 features; the caller adds the other half).

// B operand of the input MFMA (see pack_kernel's A0 layout): lane half h carries the 3-way bf16
// split of ITS input channel (h = 0: keep channel a, h = 1: keep channel b)
__device__ __forceinline__ u32x4 input_fragment(float x, int h) {
    float r1, r2, r3;
    const uint32_t xh = split_hi(x, r1), xm = split_hi(r1, r2), xl = split_hi(r2, r3);
    u32x4 b0;
    b0.x = (xh >> 16) | xm;                 // e0 = xh, e1 = xm
    b0.y = (xh >> 16) | xl;                 // e2 = xh, e3 = xl
    b0.z = (xm >> 16) | xh;                 // e4 = xm, e5 = xh
    b0.w = h ? 0x00003F80u : 0x3F803F80u;   // e6 = 1, e7 = (h == 0)
    return b0;
}

// A operand of the input MFMA for (folded weight w of this half's channel, folded bias T):
// slots [wh wh wm wh wm wl | T*], T* = (Th, Tm) for h = 0 and (Tl, 0) for h = 1; returns slot j
__device__ __forceinline__ uint32_t input_weight_slot(float w, float T, int h, int j) {
    float r1, r2, q1, q2, dummy;
    const uint32_t wh = split_hi(w, r1) >> 16, wm = split_hi(r1, r2) >> 16, wl = split_hi(r2, dummy) >> 16;
    const uint32_t Th = split_hi(T, q1) >> 16, Tm = split_hi(q1, q2) >> 16, Tl = split_hi(q2, dummy) >> 16;
    switch (j) {
        case 0: case 1: case 3: return wh;
        case 2: case 4: return wm;
        case 5: return wl;
        case 6: return h == 0 ? Th : Tl;
        default: return h == 0 ? Tm : 0u;
    }
}

// all eight slots of that A operand at once (slot j = 16-bit element j of the result)
__device__ __forceinline__ u32x4 input_weight_slots8(float w, float T, int h) {
    float r1, r2, q1, q2, dummy;
    const uint32_t wh = split_hi(w, r1) >> 16, wm = split_hi(r1, r2) >> 16, wl = split_hi(r2, dummy) >> 16;
    const uint32_t Th = split_hi(T, q1) >> 16, Tm = split_hi(q1, q2) >> 16, Tl = split_hi(q2, dummy) >> 16;
    u32x4 v;
    v.x = wh | (wh << 16);                      // j = 0, 1
    v.y = wm | (wh << 16);                      // j = 2, 3
    v.z = wm | (wl << 16);                      // j = 4, 5
    v.w = h == 0 ? (Th | (Tm << 16)) : Tl;      // j = 6, 7
    return v;
}

// feature held by accumulator register r (0..15) of M tile t in lane half h
__device__ __forceinline__ constexpr int acc_feature(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

struct FlowArgs {
    const uint8_t *packed;
    const int *meta;
    const float *film;
    const float *p_in;
    float *p_out, *p_out_pm, *sum_lv, *ps, *mus, *lvs;
    int L, B, N, mode;
    float eps;
    // optional prologue (direct mode, models.py:76-79 + :212): p_in is the NOISE and the stack starts from
    // z = p_in * exp(0.5 * lv0) + mu0, the base distribution read through its (batch, channel, point) strides -- the
    // reference's stride-0 expansions (models.py:153-158, 203-209) are never materialised; z_out (optional) receives z
    const float *base_mu, *base_lv;
    long mu_sb, mu_sc, mu_sn, lv_sb, lv_sc, lv_sn;
    float *z_out;
    // optional epilogue (training forward, csrc/flow_train.hip): moments of the output's channels xs_ka / xs_kb -- the
    // NEXT layer's kept coordinates -- per workgroup, in tstats_x_kernel's layout: xs_part[workgroup][8] doubles
    double *xs_part;
    int xs_ka, xs_kb;
    unsigned long long *prof;       // -DDPF_PROFILE builds: phase stamps; nullptr otherwise
    // 16-point-tile fragments of the same weights (csrc/flow16.hip; f16x3 only), or nullptr
    const uint8_t *packed16;
};

// ---- packed layout of the 16-point-tile kernel (csrc/flow16.hip; v_mfma_f32_16x16x32_{f16,bf16}) ----------------------
// per layer: A1 [part2][br2][t'4][s2][lane64][8 f16] (32 KiB): element j of lane (i = lane & 15, g = lane >> 4) =
//            W1[br][16 t' + i][feat(s, g, j)],  feat = 16 (2 s + (j >> 2)) + 4 g + (j & 3)
//            -- the feature register (j & 3) of M tile 2 s + (j >> 2) of the h0 accumulator holds in lane group g;
//            A0 [br2][t4][lane64][8 bf16] (8 KiB): the BN0-folded input layer's 3-way split in the K slots of lane groups
//            0 (kept channel a) and 1 (kept channel b), zeros in groups 2 and 3
constexpr int P16_A1_PART = 16384;
constexpr int P16_A1 = 2 * P16_A1_PART;
constexpr int P16_A0 = 8192;
constexpr int P16_LAYER = P16_A1 + P16_A0;


}  // namespace

// csrc-internal: defined in flow.hip, used by flow_train.hip (see there)
int flow_forward_xstats(int B, int N, int mode, int precision, const void *packed, const int *meta, const float *film,
                        const float *p_in, float *ps, float *mus, float *logvars, float flow_eps, dpf_stream_t stream,
                        double *xs_part, int xs_ka, int xs_kb, int *xs_rows);
// csrc-internal: defined in flow16.hip, called by flow.hip's dispatcher.  Returns a HIP error code, or -1000 when the
// 16-point kernel does not serve the call (the caller then takes the 32-point kernel).
int flow16_launch(const void *flow_args, hipStream_t stream);
bool flow16_serves(int n_layers, int B, int N, int precision, bool has_xs);
int flow16_pack(int n_layers, int G, const float *canon, void *packed16, hipStream_t s);
#endif  // DPF_FLOW_COMMON_H
