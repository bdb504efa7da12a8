// Micro-test (gfx950), r06 -- WHICH FORMS of the packed fp32 VALU instructions lose their low half (lanes 48-63) while another wave
// of the SIMD issues MFMAs with gaps (pk_vs_mfma_waves2.hip)?  One form per launch, four instructions of that form per chain.
//   0 v_pk_fma_f32 plain            1 v_pk_fma_f32 op_sel_hi:[1,0,1]      2 v_pk_fma_f32 op_sel:[0,1,0]
//   3 v_pk_mul_f32 plain            4 v_pk_add_f32 plain                  5 v_pk_mul_f32 op_sel_hi:[1,0]
//   6 v_pk_add_f32 V, S, V op_sel_hi:[0,1] neg (csrc/chamfer.hip's scan: a candidate pair in SGPRs against two queries)
//   7 v_pk_add_f32 V, S, V op_sel:[1,0] neg
//   8 v_fma_mixlo_f16 V, V, S, V op_sel_hi:[1,0,0] clamp  +  v_fma_mixhi_f16 V, V, S, V op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp
//     (csrc/flow.hip's relu + fp16 split, 768 of each in the headline kernel: op_sel picks the HIGH 16 bits of ONE register)
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_vs_mfma_forms.hip -o pk_vs_mfma_forms ; run: ./pk_vs_mfma_forms
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int G> __device__ __forceinline__ void gap() {
    if constexpr (G == 4) asm volatile("s_nop 4");
    if constexpr (G == 15) asm volatile("s_nop 15");
}

template <int FORM, int G>
__global__ __launch_bounds__(512) void probe(const float *__restrict__ E, int iters, int nset, unsigned long long *bad, float *sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4) {
        h8 a, b;
        for (int u = 0; u < 8; ++u) { a[u] = (_Float16)(0.01f * (lane + u)); b[u] = (_Float16)(0.02f * (lane - u)); }
        f16v c0 = {0}, c1 = {0};
        for (int it = 0; it < iters * 3; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); gap<G>();
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); gap<G>();
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1];
        return;
    }
    unsigned long long nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 5 + blockIdx.x + wave) % nset;
        const float *e = E + (size_t)set * 20 * 64 + lane;
        float x[8], w[4];
        for (int u = 0; u < 8; ++u) x[u] = e[u * 64];
        for (int u = 0; u < 4; ++u) w[u] = e[(16 + u) * 64];
        const f2 x01 = {x[0], x[1]}, x23 = {x[2], x[3]}, x45 = {x[4], x[5]}, x67 = {x[6], x[7]}, w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
        // a wave-uniform pair for the SGPR forms
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[0])));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[1])));
        const f2 sp = {s0, s1};
        f2 acc = {x[6] + 1.f, x[7] + 2.f};
        float r0 = acc.x, r1 = acc.y;
#define OPS4 : [d] "+v"(acc) : [s0] "v"(x01), [s1] "v"(x23), [s2] "v"(x45), [s3] "v"(x67), [w01] "v"(w01), [w23] "v"(w23), [sp] "s"(sp)
        if constexpr (FORM == 0) {
            asm volatile("v_pk_fma_f32 %[d], %[s0], %[w01], %[d]\n v_pk_fma_f32 %[d], %[s1], %[w23], %[d]\n"
                         "v_pk_fma_f32 %[d], %[s2], %[w01], %[d]\n v_pk_fma_f32 %[d], %[s3], %[w23], %[d]\n" OPS4);
            const float *xs[4] = {x, x + 2, x + 4, x + 6};
            for (int t = 0; t < 4; ++t) { r0 = __builtin_fmaf(xs[t][0], w[(t & 1) * 2], r0); r1 = __builtin_fmaf(xs[t][1], w[(t & 1) * 2 + 1], r1); }
        }
        if constexpr (FORM == 1) {
            asm volatile("v_pk_fma_f32 %[d], %[s0], %[w01], %[d] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %[d], %[s1], %[w23], %[d] op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %[d], %[s2], %[w01], %[d] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %[d], %[s3], %[w23], %[d] op_sel_hi:[1,0,1]\n" OPS4);
            const float *xs[4] = {x, x + 2, x + 4, x + 6};
            for (int t = 0; t < 4; ++t) { r0 = __builtin_fmaf(xs[t][0], w[(t & 1) * 2], r0); r1 = __builtin_fmaf(xs[t][1], w[(t & 1) * 2], r1); }
        }
        if constexpr (FORM == 2) {
            asm volatile("v_pk_fma_f32 %[d], %[s0], %[w01], %[d] op_sel:[0,1,0]\n v_pk_fma_f32 %[d], %[s1], %[w23], %[d] op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %[d], %[s2], %[w01], %[d] op_sel:[0,1,0]\n v_pk_fma_f32 %[d], %[s3], %[w23], %[d] op_sel:[0,1,0]\n" OPS4);
            const float *xs[4] = {x, x + 2, x + 4, x + 6};
            for (int t = 0; t < 4; ++t) { r0 = __builtin_fmaf(xs[t][0], w[(t & 1) * 2 + 1], r0); r1 = __builtin_fmaf(xs[t][1], w[(t & 1) * 2 + 1], r1); }
        }
        if constexpr (FORM == 3) {
            asm volatile("v_pk_mul_f32 %[d], %[d], %[s0]\n v_pk_mul_f32 %[d], %[d], %[w01]\n v_pk_mul_f32 %[d], %[d], %[s1]\n v_pk_mul_f32 %[d], %[d], %[w23]\n" OPS4);
            r0 = r0 * x[0]; r1 = r1 * x[1]; r0 = r0 * w[0]; r1 = r1 * w[1]; r0 = r0 * x[2]; r1 = r1 * x[3]; r0 = r0 * w[2]; r1 = r1 * w[3];
        }
        if constexpr (FORM == 4) {
            asm volatile("v_pk_add_f32 %[d], %[d], %[s0]\n v_pk_add_f32 %[d], %[d], %[w01]\n v_pk_add_f32 %[d], %[d], %[s1]\n v_pk_add_f32 %[d], %[d], %[w23]\n" OPS4);
            r0 = r0 + x[0]; r1 = r1 + x[1]; r0 = r0 + w[0]; r1 = r1 + w[1]; r0 = r0 + x[2]; r1 = r1 + x[3]; r0 = r0 + w[2]; r1 = r1 + w[3];
        }
        if constexpr (FORM == 5) {
            asm volatile("v_pk_mul_f32 %[d], %[d], %[s0] op_sel_hi:[1,0]\n v_pk_mul_f32 %[d], %[d], %[w01] op_sel_hi:[1,0]\n"
                         "v_pk_mul_f32 %[d], %[d], %[s1] op_sel_hi:[1,0]\n v_pk_mul_f32 %[d], %[d], %[w23] op_sel_hi:[1,0]\n" OPS4);
            r0 = r0 * x[0]; r1 = r1 * x[0]; r0 = r0 * w[0]; r1 = r1 * w[0]; r0 = r0 * x[2]; r1 = r1 * x[2]; r0 = r0 * w[2]; r1 = r1 * w[2];
        }
        if constexpr (FORM == 6) {     // d = S.lo - V  (both halves take the SGPR pair's low word)
            asm volatile("v_pk_add_f32 %[d], %[sp], %[d] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %[d], %[sp], %[s0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[d], %[sp], %[d] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %[d], %[sp], %[d] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n" OPS4);
            r0 = s0 - r0; r1 = s0 - r1; r0 = s0 - x[0]; r1 = s0 - x[1]; r0 = s0 - r0; r1 = s0 - r1; r0 = s0 - r0; r1 = s0 - r1;
        }
        if constexpr (FORM == 7) {     // d = S.hi - V
            asm volatile("v_pk_add_f32 %[d], %[sp], %[d] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %[d], %[sp], %[s0] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %[d], %[sp], %[d] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %[d], %[sp], %[d] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" OPS4);
            r0 = s1 - r0; r1 = s1 - r1; r0 = s1 - x[0]; r1 = s1 - x[1]; r0 = s1 - r0; r1 = s1 - r1; r0 = s1 - r0; r1 = s1 - r1;
        }
        if constexpr (FORM == 8) {
            // a = two fp16 numbers in one register; d.lo16 = f16(clamp(a.lo16 * s + c)), d.hi16 = f16(clamp(a.hi16 * s + c)), four times
            uint32_t d = 0;
            const _Float16 h0 = (_Float16)(x[0] * 0.5f), h1 = (_Float16)(x[1] * 0.5f);
            const uint32_t a = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
            float c = x[2] - 1.0f;
            uint32_t dd[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                asm volatile("v_fma_mixlo_f16 %[d], %[a], %[s], %[c] op_sel_hi:[1,0,0] clamp\n v_fma_mixhi_f16 %[d], %[a], %[s], %[c] op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n"
                             : [d] "+v"(d) : [a] "v"(a), [s] "s"(s0), [c] "v"(c));
                dd[t] = d;
                c += 0.03125f;
            }
            float cc = x[2] - 1.0f;
            unsigned long long wl = 0, wh = 0;
            for (int t = 0; t < 4; ++t) {
                const float lo = fminf(fmaxf(__builtin_fmaf((float)h0, s0, cc), 0.f), 1.f), hi = fminf(fmaxf(__builtin_fmaf((float)h1, s0, cc), 0.f), 1.f);
                const uint32_t want = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)lo) | ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)hi) << 16);
                wl += (want & 0xffffu) != (dd[t] & 0xffffu);
                wh += (want >> 16) != (dd[t] >> 16);
                cc += 0.03125f;
            }
            nlo += wl; nhi += wh;
            continue;
        }
        nlo += __float_as_uint(r0) != __float_as_uint(acc.x);
        nhi += __float_as_uint(r1) != __float_as_uint(acc.y);
    }
    if (nlo) atomicAdd(bad, nlo);
    if (nhi) atomicAdd(bad + 1, nhi);
}

static const char *NAME[] = {"v_pk_fma_f32 plain", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_mul_f32 plain", "v_pk_add_f32 plain",
                             "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_add_f32 V,S,V op_sel_hi:[0,1] neg", "v_pk_add_f32 V,S,V op_sel:[1,0] neg",
                             "v_fma_mixlo/hi_f16 (flow.hip's split)"};

template <int FORM, int G>
void run(const float *E, int nset, unsigned long long *bad, float *sink) {
    hipMemset(bad, 0, 16);
    const int iters = 20000, blocks = 512;
    hipLaunchKernelGGL((probe<FORM, G>), dim3(blocks), dim3(512), 0, 0, E, iters, nset, bad, sink);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("%-42s other wave: v_mfma; s_nop %2d : %.3g chains, wrong low halves %llu, wrong high halves %llu\n", NAME[FORM], G, (double)blocks * 256 * iters, h[0], h[1]);
}
template <int FORM> void both(const float *E, int nset, unsigned long long *bad, float *sink) { run<FORM, -1>(E, nset, bad, sink); run<FORM, 4>(E, nset, bad, sink); run<FORM, 15>(E, nset, bad, sink); }

int main() {
    const int nset = 16;
    float *E, *sink; unsigned long long *bad;
    hipMalloc(&E, nset * 20 * 64 * 4); hipMalloc(&sink, 1024 * 512 * 4); hipMalloc(&bad, 16);
    float *h = (float *)malloc(nset * 20 * 64 * 4);
    srand(5);
    for (int i = 0; i < nset * 20 * 64; ++i) h[i] = 0.75f + 0.5f * rand() / (float)RAND_MAX;
    hipMemcpy(E, h, nset * 20 * 64 * 4, hipMemcpyHostToDevice);
    both<0>(E, nset, bad, sink); both<1>(E, nset, bad, sink); both<2>(E, nset, bad, sink); both<3>(E, nset, bad, sink);
    both<4>(E, nset, bad, sink); both<5>(E, nset, bad, sink); both<6>(E, nset, bad, sink); both<7>(E, nset, bad, sink);
    both<8>(E, nset, bad, sink);
    return 0;
}
