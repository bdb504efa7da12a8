// DUAL body of the fused flow stack (included by flow.hip inside its anonymous namespace, after FlowArgs / half_sum).
//
// One wave per SIMD owns TWO 32-point tiles ("X" and "Y") and runs them as ONE statically interleaved instruction
// stream, Y eight pipeline groups (two thirds of a layer) behind X.  Why: with two waves per SIMD (flow_kernel's skewed
// ring) the hardware arbiter decides how the two tiles' MFMA chains and VALU stretches overlap, and it hides only ~55 %
// of one wave under the other; a single wave issues strictly in program order, so the order written here
// IS the order on the SIMD: every MFMA of one tile has the other tile's VALU work behind it.
//
// A tile's layer is twelve groups (one scheduling region each, paired with a group of the other tile):
//   S0      A's two input MFMAs (h0 of the logvar branch), split of A's k-step 0, accumulators <- FiLM shift D
//   S1..S4  chain A k0..k3 (6 MFMAs each) | split of the next k-step; S4 also runs B's two input MFMAs + B's first split
//   S5..S8  chain B k0..k3               | B's splits, A's output contraction in quarters
//   S9,S10  B's output contraction, sums, cross-half adds
//   S11     coupling transform, per-layer list stores, next layer's input fragment, prefetch of its first fragments
// Pairing per layer n:   phase 1  X S1..S4 (layer n)      | Y S9..S11 (layer n-1), S0 (layer n)
//                        barrier n; the staged registers of layer n+2 go to the ring slot layer n-1 just left
//                        phase 2  X S5..S8                | Y S1..S4 (layer n)
//                        phase 3  X S9..S11, S0 (n+1)     | Y S5..S8
// Weights reach LDS through registers (global_load -> VGPR -> ds_write): an LDS-DMA piece would block the only wave of
// the SIMD for 60-180 cycles per KiB; the loads of layer n+2 are issued at the top of layer n and written after barrier n.
// Ring of three one-layer buffers: after barrier n nobody reads layer n-1 (Y's S9..S11 were before it), layer n+1 (written
// after barrier n-1) is visible, and X's S11/S0 of phase 3 are its first readers.
//
// Both contraction outputs are always formed (a layer that warps one channel has zero weights for the second), and the
// transform selects by the layer's channel codes, so the body has no wave-uniform branches: every group is one basic block.

struct TileRegs {
    f32x16 acc0[2][2], acc1[2][2];     // [branch][M tile]
    u32x4 bfrag[2][2][4];              // [branch][part][k-step]
    u32x4 af[2][4];                    // [slot][part * 2 + M tile]: A fragments of one k-step
    u32x4 a0[2], b0;                   // input MFMA: weight fragments of the branch at hand, point fragment
    float pa[2][2], pb[2][2], o[2][2], lva, lvb, fa, fb;
    f32x4 cwa[2][2], cwb[2][2];          // [slot][quad & 1]: contraction weights of two pairs of quads
};
struct TilePts {
    float p0, p1, p2, s0, s1, s2;
    int nc;
};

template <bool F16, bool INV, bool LISTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flow_dual_kernel(FlowArgs a) {
    constexpr int NS = 2;
    typedef Terms<NS> TT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int LBYTES = p_layer_bytes(NS) + FILM_BYTES;
    constexpr int A0OFF = p_a0_off(NS), FILMOFF = p_layer_bytes(NS);
    constexpr int NPW = p_layer_bytes(NS) / 1024 / 4;          // KiB pieces of a layer's weights per wave
    static_assert(p_layer_bytes(NS) % 4096 == 0, "four waves share a layer's weights evenly");
    constexpr bool inverse = INV;

    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, L = a.L;
    const size_t cloud = (size_t)bi * 3 * N;
    const size_t list_stride = (size_t)a.B * 3 * N;
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float negone = -1.0f;                      // opaque to the compiler: see split_relu_f16
    asm volatile("" : "+s"(negone));
    float zero_lv = 0.0f;
    asm volatile("" : "+s"(zero_lv));
    const float v_keep = a.eps + __expf(zero_lv);
    const float k_keep = inverse ? __builtin_amdgcn_rsqf(v_keep) : __builtin_amdgcn_sqrtf(v_keep);

    // ---- weights: global -> registers -> LDS
    u32x4 stg[NPW];
    uint2 stf;
    auto fetch_layer = [&](int st) {
        const int li = inverse ? L - 1 - st : st;
        const uint8_t *src = a.packed + (size_t)li * p_layer_bytes(NS) + wave * (NPW * 1024) + lane * 16;
#pragma unroll
        for (int k = 0; k < NPW; ++k) stg[k] = *(const u32x4 *)(src + k * 1024);
        stf = *(const uint2 *)((const uint8_t *)a.film + ((size_t)li * a.B + bi) * FILM_BYTES + wave * 512 + lane * 8);
    };
    auto store_layer = [&](int st) {
        uint8_t *dst = smem + (st % 3) * LBYTES;
#pragma unroll
        for (int k = 0; k < NPW; ++k) *(u32x4 *)(dst + wave * (NPW * 1024) + k * 1024 + lane * 16) = stg[k];
        *(uint2 *)(dst + FILMOFF + wave * 512 + lane * 8) = stf;
    };
    fetch_layer(0);
    store_layer(0);
    if (L > 1) { fetch_layer(1); store_layer(1); }

    // ---- layer descriptors: one word per layer, lane l holds layers l and 64 + l (as flow_kernel)
    auto pack_meta = [&](int row) {
        const int4 m = ((const int4 *)a.meta)[min(row, L - 1)];
        return (m.x + 1) | ((m.y + 1) << 2) | ((m.z + 1) << 4) | ((m.w + 1) << 6);
    };
    const int code_lo = pack_meta(lane), code_hi = pack_meta(64 + lane);
    auto step_layer = [&](int st) { st = st < 0 ? 0 : (st >= L ? L - 1 : st); return inverse ? L - 1 - st : st; };
    auto layer_meta = [&](int l, int &k0, int &k1, int &w0, int &w1) {
        const int c = __builtin_amdgcn_readlane(l >= 64 ? code_hi : code_lo, l & 63);
        k0 = (c & 3) - 1; k1 = ((c >> 2) & 3) - 1; w0 = ((c >> 4) & 3) - 1; w1 = ((c >> 6) & 3) - 1;
    };

    // ---- the two tiles' points
    TilePts P[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = ((blockIdx.x * 4 + wave) * 2 + t) * TILE + pl;
        const int nc = n < N ? n : N - 1;      // lanes beyond the cloud mirror its last point (same values, same addresses)
        P[t].nc = nc;
        float p0 = a.p_in[cloud + nc], p1 = a.p_in[cloud + N + nc], p2 = a.p_in[cloud + 2 * (size_t)N + nc];
        if (a.base_mu != nullptr) {            // reparameterize, every op rounded as torch's (flow_kernel)
            float *pp[3] = {&p0, &p1, &p2};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float lv = a.base_lv[bi * a.lv_sb + c * a.lv_sc + nc * a.lv_sn];
                const float mu = a.base_mu[bi * a.mu_sb + c * a.mu_sc + nc * a.mu_sn];
                *pp[c] = __fadd_rn(__fmul_rn(*pp[c], expf(__fmul_rn(0.5f, lv))), mu);
            }
            if (a.z_out != nullptr && !h) {
                a.z_out[cloud + nc] = p0; a.z_out[cloud + N + nc] = p1; a.z_out[cloud + 2 * (size_t)N + nc] = p2;
            }
        }
        P[t].p0 = p0; P[t].p1 = p1; P[t].p2 = p2;
        P[t].s0 = P[t].s1 = P[t].s2 = 0.f;
    }

    // ---- pieces of a tile's layer (all indices are compile-time after unrolling: everything lives in registers).
    // Every group is cut into six SLICES, one per chain MFMA; a paired group runs slice i of X, slice i of Y, then a
    // scheduling fence, so the written order is the issued order.
    auto pin = [](auto &x) { asm volatile("" : "+v"(x)); };
    auto ld_frag1 = [&](TileRegs &T, const uint8_t *lb, int br, int ks, int slot, int j) {   // j = part * 2 + M tile
        if ((DPF_ABLATE & 8) && !(br == 0 && ks == 0)) { T.af[slot][j] = T.af[0][0] + (uint32_t)j; return; }
        T.af[slot][j] = *(const u32x4 *)(lb + (j >> 1) * P_A1_PART + (((br * 2 + (j & 1)) * 4 + ks) * 64 + lane) * 16);
    };
    auto ld_a0 = [&](TileRegs &T, const uint8_t *lb, int br, int tp) {
        T.a0[tp] = *(const u32x4 *)(lb + A0OFF + ((br * 2 + tp) * 64 + lane) * 16);
    };
    auto init_acc1 = [&](TileRegs &T, const uint8_t *lb, int br, int tp, int half) {   // accumulator starts at the FiLM shift D
        const float *film = (const float *)(lb + FILMOFF);
        if (DPF_ABLATE & 512) {
#pragma unroll
            for (int q = 8 * half; q < 8 * half + 8; ++q) T.acc1[br][tp][q] = 0.f;
            return;
        }
#pragma unroll
        for (int q = 2 * half; q < 2 * half + 2; ++q) {
            const f32x4 dv = *(const f32x4 *)(film + br * FILM_BR_FLOATS + 32 * tp + 8 * q + 4 * h);
            T.acc1[br][tp][4 * q + 0] = dv.x; T.acc1[br][tp][4 * q + 1] = dv.y;
            T.acc1[br][tp][4 * q + 2] = dv.z; T.acc1[br][tp][4 * q + 3] = dv.w;
        }
    };
    auto split_pair = [&](TileRegs &T, int br, int ks, int d) {     // relu + hi/lo split of one pair of k-step ks's B fragment
        const int t = ks >> 1, r = 8 * (ks & 1) + 2 * d;
        if (DPF_ABLATE & 1) { T.bfrag[br][0][ks][d] = f2u(T.acc0[br][t][r]); T.bfrag[br][1][ks][d] = f2u(T.acc0[br][t][r + 1]); return; }
        if (F16) {
            uint32_t hi_, lo_;
            split_relu_f16(T.acc0[br][t][r], T.acc0[br][t][r + 1], negone, hi_, lo_);
            pin(hi_); pin(lo_);
            T.bfrag[br][0][ks][d] = hi_;
            T.bfrag[br][1][ks][d] = lo_;
        } else {
            const float v0 = relu(T.acc0[br][t][r]), v1 = relu(T.acc0[br][t][r + 1]);
            float l0, l1;
            split_hi(v0, l0); split_hi(v1, l1);
            uint32_t hi_ = pack_bf16_trunc(v0, v1), lo_ = pack_bf16_rne(l0, l1);
            pin(hi_); pin(lo_);
            T.bfrag[br][0][ks][d] = hi_;
            T.bfrag[br][1][ks][d] = lo_;
        }
    };
    auto chain1 = [&](TileRegs &T, int br, int ks, int slot, int i) {   // MFMA i of the k-step: term i / 2, M tile i % 2
        const int term = i >> 1, tp = i & 1;
        if ((DPF_ABLATE & 4) && i > 0) { if (i == 1) T.acc1[br][1][0] += u2f(T.af[slot][1].x ^ T.bfrag[br][1][ks].x); return; }
        T.acc1[br][tp] = F16 ? mfma_f16(T.af[slot][TT::A[term] * 2 + tp], T.bfrag[br][TT::B[term]][ks], T.acc1[br][tp])
                             : mfma(T.af[slot][TT::A[term] * 2 + tp], T.bfrag[br][TT::B[term]][ks], T.acc1[br][tp]);
    };
    // output contraction o += W2' relu(h1 + D), eight features (register quad q of M tile tp) at a time; the weights of
    // contraction pair k (two quads) sit in slot k & 1 and are fetched while pair k - 1 is being used
    auto ld_cw = [&](TileRegs &T, const uint8_t *lb, int br, int tp, int q, int sl) {
        const float *wa = (const float *)(lb + FILMOFF) + br * FILM_BR_FLOATS + 64 + 32 * tp + 8 * q + 4 * h;
        if (DPF_ABLATE & 4096) { T.cwa[sl][q & 1] = T.cwa[sl][q & 1] * 1.5f; T.cwb[sl][q & 1] = T.cwa[sl][q & 1]; return; }
        T.cwa[sl][q & 1] = *(const f32x4 *)wa;
        T.cwb[sl][q & 1] = *(const f32x4 *)(wa + 64);
    };
    auto contract = [&](TileRegs &T, int br, int tp, int q, int sl) {
        const f32x4 wa4 = T.cwa[sl][q & 1], wb4 = T.cwb[sl][q & 1];
        if (DPF_ABLATE & 2) { T.pa[br][0] += T.acc1[br][tp][4 * q] + wa4.x; T.pb[br][0] += T.acc1[br][tp][4 * q + 1] + wb4.x; return; }
        const float r0 = relu(T.acc1[br][tp][4 * q + 0]), r1 = relu(T.acc1[br][tp][4 * q + 1]);
        const float r2 = relu(T.acc1[br][tp][4 * q + 2]), r3 = relu(T.acc1[br][tp][4 * q + 3]);
        T.pa[br][0] = __builtin_fmaf(wa4.x, r0, T.pa[br][0]); T.pa[br][1] = __builtin_fmaf(wa4.y, r1, T.pa[br][1]);
        T.pa[br][0] = __builtin_fmaf(wa4.z, r2, T.pa[br][0]); T.pa[br][1] = __builtin_fmaf(wa4.w, r3, T.pa[br][1]);
        T.pb[br][0] = __builtin_fmaf(wb4.x, r0, T.pb[br][0]); T.pb[br][1] = __builtin_fmaf(wb4.y, r1, T.pb[br][1]);
        T.pb[br][0] = __builtin_fmaf(wb4.z, r2, T.pb[br][0]); T.pb[br][1] = __builtin_fmaf(wb4.w, r3, T.pb[br][1]);
        pin(T.pa[br][0]); pin(T.pa[br][1]); pin(T.pb[br][0]); pin(T.pb[br][1]);
    };

    // ---- the twelve groups of a tile's layer, slice i of each
    // S0: A's input MFMAs (fragments prefetched by S11), split of A's k-step 0, accumulators, sums
    auto s0 = [&](TileRegs &T, const uint8_t *lb, int i) {
        if (i < 2) T.acc0[0][i] = mfma(T.a0[i], T.b0, z16);
        if (i < 4) init_acc1(T, lb, 0, i >> 1, i & 1);
        if (i >= 2) split_pair(T, 0, 0, i - 2);
        if (i == 4) {
            const float *b2 = (const float *)(lb + FILMOFF) + FILM_B2_OFF;     // the output bias starts the sums, half per lane half
#pragma unroll
            for (int br = 0; br < 2; ++br) {
                T.pa[br][0] = 0.5f * b2[br * 2]; T.pa[br][1] = 0.f;
                T.pb[br][0] = 0.5f * b2[br * 2 + 1]; T.pb[br][1] = 0.f;
            }
        }
    };
    // S1 + ks: chain A; the next k-step's fragments and split; at ks = 3 B's input MFMAs and B's first split
    auto sa = [&](TileRegs &T, const uint8_t *lb, int ks, int i) {
        if (ks == 3 && i < 2) T.acc0[1][i] = mfma(T.a0[i], T.b0, z16);
        chain1(T, 0, ks, ks & 1, i);
        if (i < 4) { if (ks < 3) ld_frag1(T, lb, 0, ks + 1, (ks + 1) & 1, i); else ld_frag1(T, lb, 1, 0, 0, i); }
        if (i >= 2) { if (ks < 3) split_pair(T, 0, ks + 1, i - 2); else split_pair(T, 1, 0, i - 2); }
        if (ks == 2 && i >= 4) ld_a0(T, lb, 1, i - 4);
        if (ks == 2 && i < 2) init_acc1(T, lb, 1, 0, i);
        if (ks == 3 && i >= 2 && i < 4) init_acc1(T, lb, 1, 1, i - 2);
        if (ks == 3 && i >= 4) ld_cw(T, lb, 0, 0, i - 4, 0);              // contraction pair 0: A, M tile 0, quads 0 1
    };
    // S5 + ks: chain B, B's next split, contraction pair ks (A: M tile ks / 2, quads 2 (ks % 2) + {0, 1})
    auto sb = [&](TileRegs &T, const uint8_t *lb, int ks, int i) {
        chain1(T, 1, ks, ks & 1, i);
        if (ks < 3 && i < 4) ld_frag1(T, lb, 1, ks + 1, (ks + 1) & 1, i);
        if (ks < 3 && (i == 0 || i == 1)) split_pair(T, 1, ks + 1, i);
        if (ks < 3 && (i == 3 || i == 4)) split_pair(T, 1, ks + 1, i - 1);
        if (i == 2) contract(T, 0, ks >> 1, 2 * (ks & 1), ks & 1);
        if (i == 5) contract(T, 0, ks >> 1, 2 * (ks & 1) + 1, ks & 1);
        if (i == 3 || i == 4) {                                            // the next pair's weights (slot parity flips)
            const int nk = ks + 1, qq = 2 * (nk & 1) + (i - 3);
            if (nk < 4) ld_cw(T, lb, 0, nk >> 1, qq, nk & 1); else ld_cw(T, lb, 1, 0, i - 3, 0);
        }
    };
    // S9: B's contraction, M tile 0 (pairs 4, 5); S10: M tile 1 (pairs 6, 7), sums, cross-half adds
    auto s9 = [&](TileRegs &T, const uint8_t *lb, int i) {
        if (i == 0 || i == 1) { contract(T, 1, 0, i, 0); ld_cw(T, lb, 1, 0, 2 + i, 1); }
        if (i == 2 || i == 3) ld_cw(T, lb, 1, 1, i - 2, 0);
        if (i == 3 || i == 4) contract(T, 1, 0, i - 1, 1);
    };
    auto s10 = [&](TileRegs &T, const uint8_t *lb, int i) {
        if (i == 0 || i == 1) { contract(T, 1, 1, i, 0); ld_cw(T, lb, 1, 1, 2 + i, 1); }
        if (i == 3 || i == 4) contract(T, 1, 1, i - 1, 1);
        if (i == 5) {
#pragma unroll
            for (int br = 0; br < 2; ++br) {
                T.o[br][0] = half_sum(T.pa[br][0] + T.pa[br][1]);
                T.o[br][1] = half_sum(T.pb[br][0] + T.pb[br][1]);
                pin(T.o[br][0]); pin(T.o[br][1]);
            }
        }
    };
    // S11: coupling transform (flows.py:96-115) of layer li with warped channels wa / wb (branch 0 = logvar, 1 = mu), the
    // per-layer lists, then the next layer's point fragment (keep channels ka / kb) and the prefetch of its first fragments
    auto s11 = [&](TileRegs &T, TilePts &Pt, int wa, int wb, int li, int ka, int kb, const uint8_t *lbn, int i) {
        if (i == 0) {
            T.lva = T.o[0][0] * __builtin_amdgcn_rcpf(1.0f + fabsf(T.o[0][0]));   // softsign, :99
            T.lvb = T.o[0][1] * __builtin_amdgcn_rcpf(1.0f + fabsf(T.o[0][1]));
            pin(T.lva); pin(T.lvb);
        }
        if (i == 1) {
            const float va = a.eps + __expf(T.lva), vb = a.eps + __expf(T.lvb);
            T.fa = inverse ? __builtin_amdgcn_rsqf(va) : __builtin_amdgcn_sqrtf(va);
            T.fb = inverse ? __builtin_amdgcn_rsqf(vb) : __builtin_amdgcn_sqrtf(vb);
            pin(T.fa); pin(T.fb);
        }
        if (i == 2) {
            float lv[3], mu[3], pn[3];
            const float pin3[3] = {Pt.p0, Pt.p1, Pt.p2};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lv[c] = c == wa ? T.lva : (c == wb ? T.lvb : 0.f);
                mu[c] = c == wa ? T.o[1][0] : (c == wb ? T.o[1][1] : 0.f);
                const float f = c == wa ? T.fa : (c == wb ? T.fb : k_keep);
                pn[c] = inverse ? (pin3[c] - mu[c]) * f : f * pin3[c] + mu[c];
            }
            Pt.p0 = pn[0]; Pt.p1 = pn[1]; Pt.p2 = pn[2];
            Pt.s0 += lv[0]; Pt.s1 += lv[1]; Pt.s2 += lv[2];
            pin(Pt.p0); pin(Pt.p1); pin(Pt.p2);
            if constexpr (LISTS) {   // per-layer lists in DIRECT order (decoders.py:61-70); the lane halves share the nine rows
                const size_t base = (size_t)li * list_stride + cloud + Pt.nc;
                float *dst[5]; float val[5];
                dst[0] = (h ? a.mus + base + 2 * (size_t)N : a.ps + base);              val[0] = h ? mu[2] : pn[0];
                dst[1] = (h ? a.lvs + base : a.ps + base + N);                          val[1] = h ? lv[0] : pn[1];
                dst[2] = (h ? a.lvs + base + N : a.ps + base + 2 * (size_t)N);          val[2] = h ? lv[1] : pn[2];
                dst[3] = (h ? a.lvs + base + 2 * (size_t)N : a.mus + base);             val[3] = h ? lv[2] : mu[0];
                dst[4] = (h ? a.lvs + base + 2 * (size_t)N : a.mus + base + N);         val[4] = h ? lv[2] : mu[1];
#pragma unroll
                for (int e = 0; e < 5; ++e) *dst[e] = val[e];
            }
        }
        if (i == 3) {
            const float xa = sel3(ka, Pt.p0, Pt.p1, Pt.p2);
            const float xb = kb < 0 ? 0.f : sel3(kb, Pt.p0, Pt.p1, Pt.p2);
            T.b0 = input_fragment(h ? xb : xa, h);
            pin(T.b0);
        }
        if (i == 4) { ld_a0(T, lbn, 0, 0); ld_a0(T, lbn, 0, 1); ld_frag1(T, lbn, 0, 0, 0, 0); }
        if (i == 5) { ld_frag1(T, lbn, 0, 0, 0, 1); ld_frag1(T, lbn, 0, 0, 0, 2); ld_frag1(T, lbn, 0, 0, 0, 3); }
    };

#define DPF_DUAL_FENCE __builtin_amdgcn_sched_barrier(0);
#ifdef DPF_PROFILE_FINE
#define DPF_TF(i) DPF_T(i)
#else
#define DPF_TF(i)
#endif
#define DPF_DUAL_GROUP(XS, YS)                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {  \
        { const int i = i_; XS; }                       \
        { const int i = i_; YS; }                       \
        DPF_DUAL_FENCE                                  \
    }

    TileRegs X, Y;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // layers 0 and 1 are in the ring

    int mk[4], mn[4];                                  // descriptors of layer n (and, inside the loop, of n-1 / n+1)
    layer_meta(step_layer(0), mk[0], mk[1], mk[2], mk[3]);
    {
        const uint8_t *lb0 = smem;
#pragma unroll
        for (int i = 3; i < 6; ++i) { s11(X, P[0], 0, 0, 0, mk[0], mk[1], lb0, i); s11(Y, P[1], 0, 0, 0, mk[0], mk[1], lb0, i); }
        DPF_DUAL_GROUP(s0(X, lb0, i), (void)0)
    }
    int mp[4] = {mk[0], mk[1], mk[2], mk[3]};          // layer n - 1
    for (int n = 0; n < L; ++n) {
        const uint8_t *lbn = smem + (n % 3) * LBYTES, *lbp = smem + ((n + 2) % 3) * LBYTES, *lbq = smem + ((n + 1) % 3) * LBYTES;
        layer_meta(step_layer(n + 1), mn[0], mn[1], mn[2], mn[3]);
        if (n + 2 < L && !(DPF_ABLATE & 2048)) fetch_layer(n + 2);
        unsigned long long tt[16];
        (void)tt;
        DPF_T(0)
        // ---- phase 1: X chain A (layer n) | Y finishes layer n-1 and starts layer n
        if (n == 0) {
            DPF_DUAL_GROUP(sa(X, lbn, 0, i), (void)0)
            DPF_DUAL_GROUP(sa(X, lbn, 1, i), (void)0)
            DPF_DUAL_GROUP(sa(X, lbn, 2, i), (void)0)
        } else {
            const int lip = step_layer(n - 1);
            DPF_DUAL_GROUP(sa(X, lbn, 0, i), s9(Y, lbp, i))
            DPF_TF(6)
            DPF_DUAL_GROUP(sa(X, lbn, 1, i), s10(Y, lbp, i))
            DPF_TF(7)
            DPF_DUAL_GROUP(sa(X, lbn, 2, i), s11(Y, P[1], mp[2], mp[3], lip, mk[0], mk[1], lbn, i))
            DPF_TF(8)
        }
        DPF_DUAL_GROUP(sa(X, lbn, 3, i), s0(Y, lbn, i))
        DPF_T(1)
        // ---- barrier n: layer n+1 is visible, layer n-1's slot is free
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        DPF_T(2)
        if (n + 2 < L && !(DPF_ABLATE & 2048)) store_layer(n + 2);
        DPF_T(3)
        // ---- phase 2: X chain B | Y chain A
        DPF_DUAL_GROUP(sb(X, lbn, 0, i), sa(Y, lbn, 0, i))
        DPF_TF(9)
        DPF_DUAL_GROUP(sb(X, lbn, 1, i), sa(Y, lbn, 1, i))
        DPF_TF(10)
        DPF_DUAL_GROUP(sb(X, lbn, 2, i), sa(Y, lbn, 2, i))
        DPF_TF(11)
        DPF_DUAL_GROUP(sb(X, lbn, 3, i), sa(Y, lbn, 3, i))
        DPF_T(4)
        // ---- phase 3: X finishes layer n and starts layer n+1 | Y chain B
        const int lin = step_layer(n);
        DPF_DUAL_GROUP(s9(X, lbn, i), sb(Y, lbn, 0, i))
        DPF_TF(12)
        DPF_DUAL_GROUP(s10(X, lbn, i), sb(Y, lbn, 1, i))
        DPF_TF(13)
        DPF_DUAL_GROUP(s11(X, P[0], mk[2], mk[3], lin, mn[0], mn[1], lbq, i), sb(Y, lbn, 2, i))
        DPF_TF(14)
        DPF_DUAL_GROUP(s0(X, lbq, i), sb(Y, lbn, 3, i))
        DPF_T(5)
#ifdef DPF_PROFILE
        if (a.prof != nullptr && lane == 0 && blockIdx.x < 2 && blockIdx.y == 0) {
            unsigned long long *o2 = a.prof + (((size_t)(blockIdx.x * 4 + wave)) * L + n) * 16;
            for (int i = 0; i < 16; ++i) o2[i] = tt[i];
        }
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) { mp[i] = mk[i]; mk[i] = mn[i]; }
    }
    // ---- Y's last layer
    {
        const uint8_t *lbl = smem + ((L - 1) % 3) * LBYTES;
#pragma unroll
        for (int i = 0; i < 6; ++i) s9(Y, lbl, i);
#pragma unroll
        for (int i = 0; i < 6; ++i) s10(Y, lbl, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) s11(Y, P[1], mp[2], mp[3], step_layer(L - 1), 0, 0, lbl, i);
    }
#undef DPF_DUAL_GROUP
#undef DPF_DUAL_FENCE

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = ((blockIdx.x * 4 + wave) * 2 + t) * TILE + pl;
        if (n < N) {
            if (!h) {
                a.p_out[cloud + n] = P[t].p0; a.p_out[cloud + N + n] = P[t].p1; a.p_out[cloud + 2 * (size_t)N + n] = P[t].p2;
                if (a.p_out_pm != nullptr) {   // point-major (B,N,3) copy for the structural losses (evaluating.py:110)
                    float *o2 = a.p_out_pm + ((size_t)bi * N + n) * 3;
                    o2[0] = P[t].p0; o2[1] = P[t].p1; o2[2] = P[t].p2;
                }
            } else if (a.sum_lv != nullptr) {
                a.sum_lv[cloud + n] = P[t].s0; a.sum_lv[cloud + N + n] = P[t].s1; a.sum_lv[cloud + 2 * (size_t)N + n] = P[t].s2;
            }
        }
    }
}

template <bool F16, bool INV>
int launch_flow_dual_dir(const FlowArgs &a, hipStream_t s) {
    const int lds = 3 * (p_layer_bytes(2) + FILM_BYTES);
    const dim3 grid((a.N + 8 * TILE - 1) / (8 * TILE), a.B), block(256);
    if (a.ps != nullptr) {
        static LdsLimit limit;
        if (hipError_t e = limit.ensure((const void *)flow_dual_kernel<F16, INV, true>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((flow_dual_kernel<F16, INV, true>), grid, block, lds, s, a);
    } else {
        static LdsLimit limit;
        if (hipError_t e = limit.ensure((const void *)flow_dual_kernel<F16, INV, false>, lds); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((flow_dual_kernel<F16, INV, false>), grid, block, lds, s, a);
    }
    return (int)hipGetLastError();
}
template <bool F16>
int launch_flow_dual(const FlowArgs &a, hipStream_t s) {
    return a.mode == DPF_MODE_INVERSE ? launch_flow_dual_dir<F16, true>(a, s) : launch_flow_dual_dir<F16, false>(a, s);
}
