// The fused L-layer coupling stack (eval mode) with ONE CONDITIONER BRANCH PER WAVE: 16 waves per workgroup, four per SIMD
// (gfx950, MI355X; f16x3 operands).  Same contract, packed weights and FiLM blocks as flow_kernel (csrc/flow.hip):
// CondRealNVPFlow3D.forward of lib/networks/flows.py:95-117 for every layer of LocalCondRNVPDecoder (decoders.py:41-72).
//
// Why (r05).  flow_kernel runs both branches of a 32-point tile in one wave: 632 instructions per wave and layer (435 VALU,
// 52 MFMA, 76 LDS, 69 scalar; rocprofv3 counters, profiles/r05_pmc_pass1-2) at 228 registers, i.e. TWO waves per SIMD.  The
// counters say what bounds it: SQ_ACTIVE_INST_ANY / wave / layer = 2 697 cycles = 632 x 4.27 -- a wave issues one instruction
// per ~4.3 cycles whatever its class, the SIMD's two waves need 5 394 of the layer's 6 155 cycles between them, and the
// matrix pipe is busy 54 %.  The issue rate is PER WAVE (tools/ubench/mfma_fill.hip, pure-VALU streams: one wave per SIMD
// 4.8 cycles per instruction, two 2.0, four 0.95): more waves issue more.  cfg-2 has two tiles per SIMD and a tile is a
// serial chain of layers, so the only way to four waves is to split a tile's work -- its two conditioner branches (logvar,
// mu) depend on the layer input only.  r02 tried "a branch per wave" with 8-wave workgroups (still two waves per SIMD:
// 52.6 us) and flow16s_kernel does it for 16-point tiles of small batches; this kernel is the 32-point form at full
// occupancy: 16 waves = 8 tiles x 2 branches share one three-slot weight ring, each wave fits 128 registers.
//
// A wave (tile, branch): input MFMAs + relu / fp16 split, the 64 x 64 contraction (24 MFMAs, fragments one k-step ahead),
// the output contraction of ITS branch, both lane halves added; its two outputs go to an LDS exchange buffer (parity-
// alternating), ONE workgroup barrier per layer -- which is also the ring's hand-over -- and both waves of a tile apply the
// coupling transform to their own copy of the points (same operations: bit-identical copies).  Waves 0-11 issue the
// LDS-DMA of layer n + 2 behind barrier n (3 pieces each), waves 12-13 its FiLM block.
#include <stdlib.h>

#include "flow_common.h"

namespace {

constexpr int SW = 16;                     // waves per workgroup
constexpr int ST = 8;                      // 32-point tiles per workgroup
constexpr int NS = 2;
constexpr int LBYTES = p_layer_bytes(NS) + FILM_BYTES;
constexpr int FILMOFF = p_layer_bytes(NS);
constexpr int XCH_OFF = 3 * LBYTES;        // exchange buffer: [parity 2][branch 2][tile ST][output 2][point 32] floats
constexpr int S_LDS = XCH_OFF + 2 * 2 * ST * 64 * 4 + 128 * 4;      // + the layer descriptors' table

__device__ __forceinline__ float half_sum(float x) {   // x(lane) + x(lane ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    return u2f(r[0]) + u2f(r[1]);
}

// one layer's packed weights (36 pieces of 1 KiB) and this cloud's FiLM block (2 pieces) straight into an LDS slot
__device__ __forceinline__ void stage32s(const FlowArgs &a, int li, int bi, uint8_t *lds, int wave, int lane_) {
    // (the lane's byte offset is made here, opaque to the optimiser: a uniform base + a 32-bit lane offset is one scalar pair and
    // one register per load; left alone the compiler hoists three 64-bit per-lane addresses out of the layer loop -- spilled, at
    // the 128 registers this kernel may use, and reloaded with a vmcnt(0) that waits for the previous DMA)
    unsigned lane = (unsigned)lane_;
    asm volatile("" : "+v"(lane));
    if (wave < 12) {
        const uint8_t *src = a.packed + (size_t)li * p_layer_bytes(NS) + wave * 3072 + lane * 16;
        uint8_t *dst = lds + wave * 3072;
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 2048, 0);
    } else if (wave < 14) {
        const int f = wave - 12;
        const uint8_t *fsrc = (const uint8_t *)a.film + ((size_t)li * a.B + bi) * FILM_BYTES + f * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((glb_void *)fsrc, (lds_void *)(lds + FILMOFF + f * 1024), 16, 0, 0);
    }
}

// One conditioner branch of one layer for one 32-point tile: the two pre-bias outputs of the branch, summed over this lane
// half's 32 features (the caller adds the halves).  The products of every accumulator element and their order are
// layer_pipe's (csrc/flow.hip).
template <bool TWO>
__device__ __forceinline__ void branch_s(const uint8_t *lb, int br, int lane, int h, u32x4 b0, float negone, float &oa, float &ob) {
    constexpr int A0OFF = p_a0_off(NS);
    typedef Terms<NS> TT;
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- h0 = BN0(W0 x) on the matrix core (3-way bf16 splits: fp32-accurate), relu + fp16 hi / lo split
    u32x4 bfrag[NS][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x16 acc0 = mfma(*(const u32x4 *)(lb + A0OFF + ((br * 2 + t) * 64 + lane) * 16), b0, z16);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {          // accumulator register r of tile t is element j = r & 7 of k-step 2 t + (r >> 3)
            const int s = 2 * t + (r >> 3), d = (r & 7) >> 1;
            uint32_t hi, lo;
            split_relu_f16(acc0[r], acc0[r + 1], negone, hi, lo);
            bfrag[0][s][d] = hi;
            bfrag[1][s][d] = lo;
        }
    }
    const float *fl = (const float *)(lb + FILMOFF) + br * FILM_BR_FLOATS;
    const float *wa = fl + 64, *wb2 = fl + 128;
    float oa0 = 0.f, oa1 = 0.f, ob0 = 0.f, ob1 = 0.f;
    // ---- h1 = W1 h0, accumulators pre-loaded with the folded FiLM shift D: both M tiles in flight (two independent MFMA
    // chains: the wave issues one every 32 cycles instead of waiting out a dependent one); the four fragments of a k-step are
    // read right in front of it -- no hand-made prefetch: the SIMD's other three waves cover the latency, and every register counts
    f32x16 acc1[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 dv = *(const f32x4 *)(fl + 32 * tp + 8 * q + 4 * h);
            acc1[tp][4 * q + 0] = dv.x; acc1[tp][4 * q + 1] = dv.y; acc1[tp][4 * q + 2] = dv.z; acc1[tp][4 * q + 3] = dv.w;
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 af[NS][2];
#pragma unroll
        for (int part = 0; part < NS; ++part)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                af[part][tp] = *(const u32x4 *)(lb + part * P_A1_PART + (((br * 2 + tp) * 4 + ks) * 64 + lane) * 16);
#pragma unroll
        for (int term = 0; term < TT::N; ++term)
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
                acc1[tp] = mfma_f16(af[TT::A[term]][tp], bfrag[TT::B[term]][ks], acc1[tp]);
    }
    // ---- o = W2' relu(h1 + D): each lane reduces its 32 features, two partial sums per output (even / odd registers)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * tp + 8 * q + 4 * h;
            const f32x4 wva = *(const f32x4 *)(wa + f0);
            const float v0 = relu(acc1[tp][4 * q + 0]), v1 = relu(acc1[tp][4 * q + 1]);
            const float v2 = relu(acc1[tp][4 * q + 2]), v3 = relu(acc1[tp][4 * q + 3]);
            oa0 = __builtin_fmaf(wva.x, v0, oa0); oa1 = __builtin_fmaf(wva.y, v1, oa1);
            oa0 = __builtin_fmaf(wva.z, v2, oa0); oa1 = __builtin_fmaf(wva.w, v3, oa1);
            if (TWO) {
                const f32x4 wvb = *(const f32x4 *)(wb2 + f0);
                ob0 = __builtin_fmaf(wvb.x, v0, ob0); ob1 = __builtin_fmaf(wvb.y, v1, ob1);
                ob0 = __builtin_fmaf(wvb.z, v2, ob0); ob1 = __builtin_fmaf(wvb.w, v3, ob1);
            }
        }
    oa = oa0 + oa1;
    ob = ob0 + ob1;
}

template <bool INV>
__global__ __launch_bounds__(SW * 64) void flow32s_kernel(FlowArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave & (ST - 1), br = wave >> 3;                  // this wave's tile and conditioner branch (0 logvar, 1 mu)
    const int N = a.N, L = a.L;
    const int n = (blockIdx.x * ST + tile) * TILE + pl;
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
        if (a.z_out != nullptr && valid && !h && br == 0) {
            a.z_out[cloud + n] = p0; a.z_out[cloud + N + n] = p1; a.z_out[cloud + 2 * (size_t)N + n] = p2;
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;   // running sum of logvar per channel
    constexpr bool inverse = INV;
    const size_t list_stride = (size_t)a.B * 3 * N;
    float negone = -1.0f;                      // opaque to the compiler: see split_relu_f16
    asm volatile("" : "+s"(negone));
    float zero_lv = 0.0f;                      // a kept channel's logvar, opaque so that its factor is computed, not folded
    asm volatile("" : "+s"(zero_lv));
    const float v_keep = a.eps + __expf(zero_lv);
    const float k_keep = inverse ? __builtin_amdgcn_rsqf(v_keep) : __builtin_amdgcn_sqrtf(v_keep);
    const int lfirst = inverse ? L - 1 : 0;
    auto stage_step = [&](int st) {                      // the layer of step st into ring slot st % 3
        if (st < L) stage32s(a, inverse ? L - 1 - st : st, bi, smem + (st % 3) * LBYTES, wave, lane);
    };
    stage_step(0); stage_step(1);
    // layer descriptors: one packed word per layer in an LDS table (a broadcast ds_read + v_readfirstlane per layer).  A load
    // from global memory inside the loop would put a vector-memory wait at the top of every layer, and vmcnt retires in order:
    // it would wait for the weight DMA issued just before (flow_kernel keeps the words in two registers per lane instead;
    // this kernel has none to spare)
    int *metatab = (int *)(smem + XCH_OFF + 2 * 2 * ST * 64 * 4);
    if (threadIdx.x < 128) {
        const int4 m = ((const int4 *)a.meta)[min((int)threadIdx.x, L - 1)];
        metatab[threadIdx.x] = (m.x + 1) | ((m.y + 1) << 2) | ((m.z + 1) << 4) | ((m.w + 1) << 6);
    }
    auto layer_meta = [&](int l, int &k0, int &k1, int &w0, int &w1) {      // L <= 128 (checked by the launcher)
        const int c = __builtin_amdgcn_readfirstlane(metatab[l]);
        k0 = (c & 3) - 1; k1 = ((c >> 2) & 3) - 1; w0 = ((c >> 4) & 3) - 1; w1 = ((c >> 6) & 3) - 1;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first layers' weights (and everything above) have landed
    __syncthreads();
    int ka, kb, wa, wb;
    layer_meta(lfirst, ka, kb, wa, wb);
    float *xch = (float *)(smem + XCH_OFF);

    for (int step = 0; step < L; ++step) {
        const int li = inverse ? L - 1 - step : step;
        const int ln = inverse ? (li > 0 ? li - 1 : 0) : (li + 1 < L ? li + 1 : li);
        const uint8_t *lb = smem + (step % 3) * LBYTES;
        int nka, nkb, nwa, nwb;
        layer_meta(ln, nka, nkb, nwa, nwb);
        const float xa = sel3(ka, p0, p1, p2);
        const float xb = kb < 0 ? 0.f : sel3(kb, p0, p1, p2);
        const u32x4 b0 = input_fragment(h ? xb : xa, h);
        float oa, ob;
        if (wb < 0) branch_s<false>(lb, br, lane, h, b0, negone, oa, ob);          // the layer warps one channel
        else branch_s<true>(lb, br, lane, h, b0, negone, oa, ob);
        const float *b2 = (const float *)(lb + FILMOFF) + FILM_B2_OFF;
        oa = half_sum(oa) + b2[br * 2 + 0];
        ob = half_sum(ob) + b2[br * 2 + 1];
        // ---- the branches meet: this wave's two outputs out, the partner wave's in.  ONE barrier per layer; it also hands the
        // ring on: behind it nobody reads layer step - 1's slot any more, and layer step + 1's DMA (issued a layer ago, waited
        // for here by its issuers) is published by it
        float *mine = xch + (size_t)((((step & 1) * 2 + br) * ST + tile) * 64);
        if (!h) { mine[pl] = oa; mine[32 + pl] = ob; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage_step(step + 2);
        const float *other = xch + (size_t)((((step & 1) * 2 + (1 - br)) * ST + tile) * 64);
        const float qa = other[pl], qb = other[32 + pl];
        const float lo_a = br ? qa : oa, lo_b = br ? qb : ob;              // the logvar branch's outputs
        const float mu_a = br ? oa : qa, mu_b = br ? ob : qb;              // the mu branch's
        // ---- coupling transform (flows.py:96-115), as flow_kernel: only the warped channels go through softsign / exp / sqrt
        float lva, lvb = 0.f, fa, fb = k_keep;
        lva = lo_a * __builtin_amdgcn_rcpf(1.0f + fabsf(lo_a));   // softsign, :99
        const float va = a.eps + __expf(lva);
        fa = inverse ? __builtin_amdgcn_rsqf(va) : __builtin_amdgcn_sqrtf(va);
        if (wb >= 0) {
            lvb = lo_b * __builtin_amdgcn_rcpf(1.0f + fabsf(lo_b));
            const float vb = a.eps + __expf(lvb);
            fb = inverse ? __builtin_amdgcn_rsqf(vb) : __builtin_amdgcn_sqrtf(vb);
        }
        float lv[3], mu[3], pn[3];
        const float pin[3] = {p0, p1, p2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lv[c] = c == wa ? lva : (c == wb ? lvb : 0.f);
            mu[c] = c == wa ? mu_a : (c == wb ? mu_b : 0.f);
            const float f = c == wa ? fa : (c == wb ? fb : k_keep);
            pn[c] = inverse ? (pin[c] - mu[c]) * f : f * pin[c] + mu[c];
        }
        p0 = pn[0]; p1 = pn[1]; p2 = pn[2];
        s0 += lv[0]; s1 += lv[1]; s2 += lv[2];
        if (a.ps != nullptr) {   // per-layer lists in DIRECT order (decoders.py:61-70): the logvar wave writes ps and logvars, the mu wave mus
            // (the point index is made again from the thread index: 64-bit addresses carried across the loop are registers this
            // kernel does not have)
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int n2 = (blockIdx.x * ST + ((t2 >> 6) & (ST - 1))) * TILE + (t2 & 31);
            const size_t base = (size_t)li * list_stride + (size_t)bi * 3 * N + n2;
            if (n2 >= N) {
            } else if (br == 0) {
                float *d = h ? a.lvs + base : a.ps + base;
                d[0] = h ? lv[0] : pn[0]; d[N] = h ? lv[1] : pn[1]; d[2 * (size_t)N] = h ? lv[2] : pn[2];
            } else if (!h) {
                float *d = a.mus + base;
                d[0] = mu[0]; d[N] = mu[1]; d[2 * (size_t)N] = mu[2];
            }
        }
        ka = nka; kb = nkb; wa = nwa; wb = nwb;
    }
    int t3 = threadIdx.x;                  // (as above: the output addresses are made here, not carried through the loop)
    asm volatile("" : "+v"(t3));
    const int n3 = (blockIdx.x * ST + ((t3 >> 6) & (ST - 1))) * TILE + (t3 & 31);
    const size_t cloud3 = (size_t)bi * 3 * N;
    if (n3 < N && br == 0) {
        if (!((t3 >> 5) & 1)) {
            a.p_out[cloud3 + n3] = p0; a.p_out[cloud3 + N + n3] = p1; a.p_out[cloud3 + 2 * (size_t)N + n3] = p2;
            if (a.p_out_pm != nullptr) {   // point-major (B,N,3) copy for the structural losses (evaluating.py:110)
                float *o2 = a.p_out_pm + ((size_t)bi * N + n3) * 3;
                o2[0] = p0; o2[1] = p1; o2[2] = p2;
            }
        } else if (a.sum_lv != nullptr) {
            a.sum_lv[cloud3 + n3] = s0; a.sum_lv[cloud3 + N + n3] = s1; a.sum_lv[cloud3 + 2 * (size_t)N + n3] = s2;
        }
    }
}

int g_split_mode = getenv("DPF_FLOW_SPLIT32") ? atoi(getenv("DPF_FLOW_SPLIT32")) : -1;
long g_split_launches = 0;

template <bool INV>
int launch32s(const FlowArgs &a, hipStream_t s) {
    static LdsLimit limit;
    if (hipError_t e = limit.ensure((const void *)flow32s_kernel<INV>, S_LDS); e != hipSuccess) return (int)e;
    const dim3 grid((a.N + TILE * ST - 1) / (TILE * ST), a.B), block(SW * 64);
    hipLaunchKernelGGL((flow32s_kernel<INV>), grid, block, S_LDS, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// mode: -1 = by size (default; env DPF_FLOW_SPLIT32), 0 = never, 1 = whenever the precision allows.  Returns the previous mode.
extern "C" int dpf_flow_set_split32(int mode) {
    const int old = g_split_mode;
    g_split_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return old;
}
extern "C" long dpf_flow_split32_launches(void) { return g_split_launches; }

// f16x3, no training epilogue; by default where every CU gets a workgroup of 256 points (below that the 16-point-tile kernels
// or flow_kernel's smaller workgroups fill the chip better)
bool flow32s_serves(int n_layers, int B, int N, int precision, bool has_xs) {
    if (precision != DPF_PREC_F16X3 || has_xs || g_split_mode == 0 || n_layers > 128 || B > 65535) return false;
    if (g_split_mode == 1) return true;
    return (long)B * ((N + 255) / 256) >= 224;
}

int flow32s_launch(const void *flow_args, hipStream_t stream) {
    const FlowArgs &a = *(const FlowArgs *)flow_args;
    ++g_split_launches;
    return a.mode == DPF_MODE_INVERSE ? launch32s<true>(a, stream) : launch32s<false>(a, stream);
}
