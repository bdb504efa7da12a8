// r03 experiment, NOT part of the library (see tools/experiments/README.md): pass 2 of the training backward with one
// conditioner branch per wave (16 waves per workgroup, 4 per SIMD).  Parity-green, 28.6 us against 25.3 us for tbwd2_kernel.
// To try it again: paste into csrc/flow_train.hip after tbwd2_kernel and launch it with dim3(TW2 * 64).
// Pass 2 with ONE BRANCH PER WAVE (r03): 16 waves per workgroup -- waves 0-7 run the logvar branch of the workgroup's
// eight 32-point tiles, waves 8-15 the mu branch of the same tiles -- so a SIMD holds FOUR independent instruction streams
// instead of two.  The per-wave stream of tbwd2_kernel is latency-bound (a single wave per SIMD needs 23 us for ~3 000
// instructions that would issue in 6: dependent MFMA -> VALU -> LDS chains), and at B*N = 65 536 points there are only two
// 32-point tiles per SIMD; halving the stream and doubling the streams hides that latency.  Same arithmetic per (tile,
// branch) as tbwd2_kernel -- the part2 rows are bit-identical -- except u_k, whose two branch halves are now added as
// (sum over branch 0) + (sum over branch 1) through LDS instead of in one running register.  128 VGPRs per wave, the same
// LDS (the 64 KB of reduction slots are 16 x 4 KB: four rounds of 16 accumulator registers instead of two of 32).
constexpr int TW2 = 2 * TW;
template <int NS, bool F16 = false>
__global__ __launch_bounds__(TW2 * 64) void tbwd2s_kernel(TArgs a, const float *__restrict__ pcs, double count, float *__restrict__ dcanon_l,
                                                           const float *__restrict__ dout,
                                                           float *__restrict__ ubuf, float *__restrict__ part2) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int L_FILM = l_film(NS), L_FILMB = l_filmb(NS), L_RED = l_red(NS);
    float *cf = (float *)(smem + L_RED) + 256;                             // c_fk [2 br][2][64]   (first 1 KB of the region: spare)
    float *redw = (float *)(smem + L_RED + 4096);                          // 16 per-wave slots of 4 KB; their head is the wave's per-point scratch before the reduction
    const int bi = blockIdx.y, lane = threadIdx.x & 63, h = lane >> 5, pl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tw = wave & (TW - 1), br = wave >> 3;                        // tile of the workgroup, branch of this wave
    int h4 = 4 * h;
    asm volatile("" : "+v"(h4));
    // ---- prologue loads before the weight DMA (threads 0..511, as tbwd2_kernel)
    const bool lo512 = threadIdx.x < TW * 64;
    const int mq = threadIdx.x & 127, mbr = mq >> 6, mf = mq & 63, mg4 = (threadIdx.x >> 7) & 3;
    float m_av[8], m_q3[8], m_q2[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int b = mg4 + 4 * jj;
        const bool ok = lo512 && b < a.B;
        const float *qq = pcs + (size_t)(ok ? b : 0) * 520 + mbr * 256;
        m_av[jj] = ok ? a.filmb_l[(size_t)b * FB_CLOUD + mbr * FB_BR + mf] : 0.f;
        m_q3[jj] = ok ? qq[3 * 64 + mf] : 0.f;
        m_q2[jj] = ok ? qq[2 * 64 + mf] : 0.f;
    }
    float cf_w = 0.f, cf_r = 0.f, cf_g = 0.f, w2_v = 0.f;
    if (threadIdx.x < 256) {                                               // c_fk = W0[f][k] * rstd0_f * gamma0_f
        const int b2 = threadIdx.x >> 7, k = (threadIdx.x >> 6) & 1, f = threadIdx.x & 63;
        const float *cb = a.tcanon_l + b2 * T_BR;
        const int nk = a.kb >= 0 ? 2 : 1;
        cf_w = k < nk ? cb[T_W0 + f * nk + k] : 0.f; cf_r = a.stats_l[b2 * ST_BR + 64 + f]; cf_g = cb[T_G0 + f];
    } else if (lo512) {
        const int i = threadIdx.x - 256;
        w2_v = a.tcanon_l[(i >> 7) * T_BR + T_W2 + (i & 127)];
    }
    const int N = a.N, n = (blockIdx.x * TW + tw) * TILE + pl;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const float *pc = a.p_in + (size_t)bi * 3 * N;
    const float xa = pc[(size_t)a.ka * N + nc], xb = a.kb >= 0 ? pc[(size_t)a.kb * N + nc] : 0.f;
    const float *dq = dout + ((size_t)bi * 4 + 2 * br) * N + nc;
    const float doa_pt = dq[0], dob_pt = dq[N];                            // d(o) of THIS wave's branch
    asm volatile("" ::: "memory");
    for (int c = wave; c * 1024 < pt_bytes(NS); c += TW2)
        __builtin_amdgcn_global_load_lds((glb_void *)(a.packed_l + c * 1024 + lane * 16), (lds_void *)(smem + L_PACK + c * 1024), 16, 0, 0);
    if (wave < 2)
        __builtin_amdgcn_global_load_lds((glb_void *)((const uint8_t *)(a.film_l + (size_t)bi * 512) + wave * 1024 + lane * 16),
                                         (lds_void *)(smem + L_FILM + wave * 1024), 16, 0, 0);
    else if (wave < 4)
        __builtin_amdgcn_global_load_lds((glb_void *)((const uint8_t *)(a.filmb_l + (size_t)bi * FB_CLOUD) + (wave - 2) * 1024 + lane * 16),
                                         (lds_void *)(smem + L_FILMB + (wave - 2) * 1024), 16, 0, 0);
    const u32x4 b0 = input_fragment(h ? xb : xa, h);
    const float *film = (const float *)(smem + L_FILM);
    const float *filmb = (const float *)(smem + L_FILMB);
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    float *s12s = cf + 256;                                                // [2 br][2][64] BN1 backward means
    float *w2s = s12s + 256;                                               // [2 br][2][64] raw sd2.weight
    if (threadIdx.x < 256) cf[threadIdx.x] = cf_w * cf_r * cf_g;
    else if (lo512) w2s[threadIdx.x - 256] = w2_v;
    {   // BN1-backward means (see tbwd2_kernel): same sums in the same order, threads 0..511
        double (*acc)[5][128] = (double (*)[5][128])redw;
        const int q = mq, br_ = mbr, f_ = mf, g4 = mg4;
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        double S1 = 0, S2 = 0, w2a = 0, w2b = 0, bb = 0;
        if (lo512) {
            for (int b0c = 0; b0c < a.B; b0c += 32) {
                if (b0c > 0) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int b = b0c + g4 + 4 * jj;
                        const bool ok = b < a.B;
                        const float *qq = pcs + (size_t)(ok ? b : 0) * 520 + br_ * 256;
                        m_av[jj] = ok ? a.filmb_l[(size_t)b * FB_CLOUD + br_ * FB_BR + f_] : 0.f;
                        m_q3[jj] = ok ? qq[3 * 64 + f_] : 0.f;
                        m_q2[jj] = ok ? qq[2 * 64 + f_] : 0.f;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    S1 += (double)m_av[jj] * m_q3[jj];
                    S2 += (double)m_av[jj] * m_q2[jj];
                }
                if (first)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int b = b0c + g4 + 4 * jj;
                        if (b >= a.B) break;
                        const float *qq = pcs + (size_t)b * 520 + br_ * 256;
                        w2a += qq[0 * 64 + f_]; w2b += qq[1 * 64 + f_];
                        if (f_ < 2) bb += pcs[(size_t)b * 520 + 512 + br_ * 2 + f_];
                    }
            }
            acc[g4][0][q] = S1; acc[g4][1][q] = S2; acc[g4][2][q] = w2a; acc[g4][3][q] = w2b; acc[g4][4][q] = bb;
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            double T1 = 0, T2 = 0, v2a = 0, v2b = 0, vb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { T1 += acc[k][0][q]; T2 += acc[k][1][q]; v2a += acc[k][2][q]; v2b += acc[k][3][q]; vb += acc[k][4][q]; }
            s12s[(br_ * 2 + 0) * 64 + f_] = (float)(T1 / count);
            s12s[(br_ * 2 + 1) * 64 + f_] = (float)(T2 / count);
            if (first) {
                dcanon_l[br_ * T_BR + T_W2 + f_] = (float)v2a;
                dcanon_l[br_ * T_BR + T_W2 + 64 + f_] = (float)v2b;
                if (f_ < 4) dcanon_l[br_ * T_BR + T_B2 + f_] = (float)vb;
            }
        }
    }
    float ua = 0.f, ub = 0.f;
    u32x4 eye[2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int p0 = ((2 * d) & 3) + 8 * (2 * j2 + ((2 * d) >> 2)) + 4 * h, p1 = p0 + 1;
            eye[j2][d] = (pl == p0 ? 0x3F80u : 0u) | (pl == p1 ? 0x3F800000u : 0u);
        }
    float *pts = redw + wave * 1024;                                       // per-wave scratch (free until the reduction): [4][32] per-point values
    const int tile0 = (blockIdx.x * TW + tw) * TILE;
    __syncthreads();                                                       // staging landed, means published, the means' scratch is free
    if (!h) { pts[pl] = doa_pt; pts[32 + pl] = dob_pt; pts[64 + pl] = xa; pts[96 + pl] = xb; }
    // ---- forward recomputation (as tbwd2_kernel)
    f32x16 pre[2];
    {
        u32x4 bf[NS][4];
        f32x16 h0a[2];                                                     // dead after the split: the ReLU mask below comes from a second input MFMA (2 MFMAs) instead of 32 live registers
        input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, h0a);
        split_fragment<true, NS, false, F16>(h0a, bf, a.negone);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float dsh = film[br * FILM_BR_FLOATS + 32 * t + pl];
#pragma unroll
            for (int r = 0; r < 16; ++r) pre[t][r] = dsh;
        }
        chain_mfma_swapped<NS, F16>(smem + L_PACK + PT_A1, br, lane, bf, pre);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32x4 xh[2][2], xl[2][2];
    {
        f32x4 doa4[4], dob4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { doa4[q] = *(const f32x4 *)(pts + 8 * q + h4); dob4[q] = *(const f32x4 *)(pts + 32 + 8 * q + h4); }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int fo = 32 * t + pl;
            const float av = filmb[br * FB_BR + 0 * 64 + fo], rstd1 = filmb[br * FB_BR + 2 * 64 + fo], ca = filmb[br * FB_BR + 3 * 64 + fo];
            const float w2a = w2s[br * 128 + fo], w2b = w2s[br * 128 + 64 + fo];
            const float m1 = s12s[br * 128 + fo], m2 = s12s[br * 128 + 64 + fo];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g2 = dh2a_of(pre[t][r], w2a, w2b, doa4[r >> 2][r & 3], dob4[r >> 2][r & 3]);
                const float h1n = pre[t][r] * rstd1 - ca;
                const float v = rstd1 * (av * g2 - m1 - h1n * m2);
                pre[t][r] = tile0 + (r & 3) + 8 * (r >> 2) + 4 * h < N ? v : 0.f;
            }
        }
        kfrags_from_swapped<false>(pre, xh, xl);
    }
    u32x4 bg[2][4];
    {
        f32x16 dn[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dn[t] = zero16();
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                dn[t] = mfma(xh[t][j2], eye[j2], dn[t]);
                dn[t] = mfma(xl[t][j2], eye[j2], dn[t]);
            }
        }
        split_fragment<false, 2, true>(dn, bg);
    }
    {
        f32x16 dh0a[2] = {zero16(), zero16()};
        chain_mfma<2>(smem + L_PACK + pt_a1t(NS), br, lane, bg, dh0a);
        {
            f32x16 h0m[2];
            input_mfma(smem + L_PACK + pt_a0(NS), br, lane, b0, h0m);      // the same two MFMAs as above: the same bits
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh0a[t][r] = h0m[t][r] > 0.f ? dh0a[t][r] : 0.f;
        }
        int h4u = h4;
        asm volatile("" : "+v"(h4u) : "v"(dh0a[1][15]));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const float *c0 = cf + br * 128 + h4u + 32 * t;
#pragma unroll
                for (int q = 2 * rh; q < 2 * rh + 2; ++q) {
                    const f32x4 ca = *(const f32x4 *)(c0 + 8 * q), cb = *(const f32x4 *)(c0 + 64 + 8 * q);
                    ua += ca.x * dh0a[t][4 * q + 0]; ub += cb.x * dh0a[t][4 * q + 0];
                    ua += ca.y * dh0a[t][4 * q + 1]; ub += cb.y * dh0a[t][4 * q + 1];
                    ua += ca.z * dh0a[t][4 * q + 2]; ub += cb.z * dh0a[t][4 * q + 2];
                    ua += ca.w * dh0a[t][4 * q + 3]; ub += cb.w * dh0a[t][4 * q + 3];
                }
                asm volatile("" : "+v"(h4u) : "v"(ua), "v"(ub));
            }
    }
    float rsum[2][4];
    f32x16 h0s[2];
    {
        input_mfma_swapped(smem + L_PACK + pt_a0(NS), br, lane, b0, h0s);
        f32x16 dh0s[2] = {zero16(), zero16()};
        chain_mfma_swapped<2>(smem + L_PACK + pt_a1t(NS), br, lane, bg, dh0s);
        f32x16 h0n[2];
        input_mfma_swapped(smem + L_PACK + pt_a0n(NS), br, lane, b0, h0n);
        f32x4 xa4[4], xb4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { xa4[q] = *(const f32x4 *)(pts + 64 + 8 * q + h4); xb4[q] = *(const f32x4 *)(pts + 96 + 8 * q + h4); }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = h0s[t][r] > 0.f ? dh0s[t][r] : 0.f;
                s0 += d * h0n[t][r]; s1 += d; s2 += d * xa4[r >> 2][r & 3]; s3 += d * xb4[r >> 2][r & 3];
            }
            rsum[t][0] = s0 + __shfl_xor(s0, 32); rsum[t][1] = s1 + __shfl_xor(s1, 32);
            rsum[t][2] = s2 + __shfl_xor(s2, 32); rsum[t][3] = s3 + __shfl_xor(s3, 32);
        }
    }
    // ---- dW1 = dh1 h0^T on the matrix cores, one output-feature tile (mt) at a time: its 12 MFMAs, then its two reduction
    // rounds (nt) of 16 accumulator registers.  Waves 0-7 feed branch 0's row, waves 8-15 branch 1's; thread (half = its own
    // wave's branch, quad e < 256) sums one quad over the eight waves of its branch, in wave order -- the order tbwd2_kernel
    // adds them in.
    u32x4 yh[2][2], yl[2][2];
    kfrags_from_swapped<true>(h0s, yh, yl);
    const int half = threadIdx.x >> 9, e = threadIdx.x & 511;
    float *o = part2 + (blk * 2 + half) * P2_J;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        f32x16 dw[2] = {zero16(), zero16()};                              // [fi tile]
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                dw[nt] = mfma(xl[mt][j2], yh[nt][j2], dw[nt]);
                dw[nt] = mfma(xh[mt][j2], yl[nt][j2], dw[nt]);
                dw[nt] = mfma(xh[mt][j2], yh[nt][j2], dw[nt]);
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            __syncthreads();                                               // the scratch / the previous round is free
            f32x4 *slot = (f32x4 *)(redw + wave * 1024) + lane;            // quad (g, lane) at [g * 64 + lane]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {dw[nt][4 * g + 0], dw[nt][4 * g + 1], dw[nt][4 * g + 2], dw[nt][4 * g + 3]};
                slot[g * 64] = v;
            }
            __syncthreads();
            if (e < 256) {
                const float *base = redw + half * TW * 1024;
                f32x4 t = *((const f32x4 *)base + e);
#pragma unroll
                for (int w = 1; w < TW; ++w) {
                    const f32x4 v = *((const f32x4 *)(base + w * 1024) + e);
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                const int ln = e & 63, g = e >> 6;
                const int fo = 32 * mt + 8 * g + 4 * (ln >> 5), fi = 32 * nt + (ln & 31);
                o[128 + (fo + 0) * 64 + fi] = t.x;
                o[128 + (fo + 1) * 64 + fi] = t.y;
                o[128 + (fo + 2) * 64 + fi] = t.z;
                o[128 + (fo + 3) * 64 + fi] = t.w;
            }
        }
    }
    __syncthreads();
    if (!h) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) redw[wave * 256 + k * 64 + 32 * t + pl] = rsum[t][k];
    }
    ua += __shfl_xor(ua, 32); ub += __shfl_xor(ub, 32);
    float *uks = redw + TW2 * 256;                                         // [8 tiles][2][32]: branch 1's share of u_k
    if (br == 1 && h == 0) { uks[tw * 64 + pl] = ua; uks[tw * 64 + 32 + pl] = ub; }
    __syncthreads();
    if (e < 256) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < TW; ++w) t += redw[(half * TW + w) * 256 + e];
        o[e < 128 ? e : 4224 - 128 + e] = t;
    }
    if (br == 0 && valid && h == 0) {
        ubuf[((size_t)bi * 2 + 0) * N + n] = ua + uks[tw * 64 + pl];
        ubuf[((size_t)bi * 2 + 1) * N + n] = ub + uks[tw * 64 + 32 + pl];
    }
}

