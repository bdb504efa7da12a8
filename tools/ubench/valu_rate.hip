// Micro-benchmark: issue rate of VALU / MFMA instruction streams on gfx950, per SIMD,
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void k(float *out, int iters, float seed, unsigned long long *ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i, seed - i};
    f32x16 acc = {0};
    f32x16 acc2 = {0};
    f32x16 acc3 = {0};
    f32x16 acc4 = {0};
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    uint32_t u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {        // 128 independent-ish v_fma_f32 (8 chains)
            REP16(for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);)
        } else if (KIND == 1) { // 128 v_pk_fma_f32
            REP16(for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{1.0001f, 1.0002f}, f2{0.5f, 0.25f});)
        } else if (KIND == 2) { // 128 v_max_i32 / and mix
            REP16(for (int i = 0; i < 8; ++i) u[i] = (uint32_t)max((int)(u[i] + 0x9E3779B9u), 7);)
        } else if (KIND == 3) { // 16 MFMA only (2 accumulators)
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
            }
        } else if (KIND == 4) { // 16 MFMA + 128 fma interleaved by the compiler
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            }
        } else if (KIND == 6) { // 16 MFMA + 128 integer VALU (and/max)
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                for (int i = 0; i < 8; ++i) u[i] = (uint32_t)max((int)(u[i] ^ 0x5bd1e995u), (int)i);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
                for (int i = 0; i < 8; ++i) u[i] = (uint32_t)max((int)(u[i] ^ 0x5bd1e995u), (int)i);
            }
        } else if (KIND == 7) { // 256 integer VALU only (xor + max)
            REP16(for (int i = 0; i < 8; ++i) u[i] = (uint32_t)max((int)(u[i] ^ 0x5bd1e995u), (int)i);)
        } else if (KIND == 8) { // 16 MFMA on 4 accumulators + 128 fma
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc3, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                acc4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc4, 0, 0, 0);
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            }
        } else if (KIND == 9) { // wave-specialised: waves 0-3 of each 8 do 16 MFMA, waves 4-7 do 128 fma
            if (((threadIdx.x >> 6) >> 2) & 1) {
                REP16(for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);)
            } else {
                for (int r = 0; r < 8; ++r) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
                }
            }
        } else if (KIND == 10) { // 16 MFMA (4 acc) then 128 fma, NOT interleaved in program order
            for (int r = 0; r < 4; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc3, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc4, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            REP16(for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);)
            __builtin_amdgcn_sched_barrier(0);
        } else if (KIND == 5) { // v_perm + cvt_pk mix (128)
            REP16(for (int i = 0; i < 8; ++i) u[i] = __builtin_amdgcn_perm(u[i], u[(i + 1) & 7], 0x07060302u);)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i] + acc3[i] + acc4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (ticks && blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int KIND>
void run(const char *name, int per_iter) {
    float *d;
    hipMalloc(&d, 256 * 16 * 256 * 4 * 4);
    unsigned long long *dt; hipMalloc(&dt, 8); unsigned long long ht = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wpc : {4, 8, 16, 32}) {          // waves per CU -> waves per SIMD = wpc/4
        dim3 grid(256 * (wpc >= 16 ? wpc / 16 : 1)), block(64 * (wpc >= 16 ? 16 : wpc));
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 10, 1.0f, (unsigned long long *)nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f, dt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * per_iter * (wpc / 4.0);
        hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
        printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD; s_memtime %.3f ticks/ns; %.2f ticks per instr\n", name, wpc / 4, ms,
               ms * 1e6 / inst_per_simd, (double)ht / (ms * 1e6), (double)ht / inst_per_simd);
    }
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 128);
    run<1>("v_pk_fma_f32", 128);
    run<2>("v_add+v_max_i32", 256);
    run<5>("v_perm_b32", 128);
    run<3>("mfma32x32x16", 16);
    run<4>("mfma + 8 fma each", 16 + 128);
    run<8>("mfma(4 acc) + 8 fma", 16 + 128);
    run<7>("xor+max_i32 only", 256);
    run<6>("mfma + 8x(xor,max)", 16 + 256);
    run<9>("specialised waves (per pair: 16 mfma | 128 fma)", 72);
    run<10>("16 mfma then 128 fma", 16 + 128);
    return 0;
}
