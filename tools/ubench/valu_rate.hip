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
        } else if (KIND == 11) { // 128 v_exp_f32 (8 chains)
            REP16(for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);)
        } else if (KIND == 12) { // the approx-EMD candidate step x128: 3 sub, mul, 2 fma, mul, exp, fma (9 VALU)
            auto step = [&]() {
                asm volatile("" : "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));      // opaque: no CSE across candidates
                const float dx = seed - a[1], dy = seed - a[2], dz = seed - a[3];
                const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                a[0] = __builtin_fmaf(__builtin_amdgcn_exp2f(d * seed), seed, a[0]);
            };
            REP16(for (int i = 0; i < 8; ++i) step();)
        } else if (KIND >= 13 && KIND <= 19) { // single-opcode streams through inline asm, 8 chains x 16
            REP16(for (int i = 0; i < 8; ++i) {
                if (KIND == 13) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 14) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
                if (KIND == 15) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 16) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 17) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(a[(i + 1) & 7]));
                if (KIND == 18) asm volatile("v_fma_f32 %0, %1, 1.0, -%0" : "+v"(a[i]) : "s"(seed));
                if (KIND == 19) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            })
        } else if (KIND >= 20 && KIND <= 29) { // packed / perm / cvt streams, VGPR-only vs SGPR operands
            const unsigned long long sp = ((unsigned long long)__float_as_uint(seed) << 32) | __float_as_uint(seed);
            REP16(for (int i = 0; i < 8; ++i) {
                if (KIND == 20) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (KIND == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(iters));
                if (KIND == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (KIND == 23) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sp), "v"(p[(i + 1) & 7]));
                if (KIND == 24) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (KIND == 25) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "s"(sp));
                if (KIND == 26) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
                if (KIND == 27) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 28) asm volatile("v_max_i32_e32 %0, 0, %0" : "+v"(u[i]));
                if (KIND == 29) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            })
        } else if (KIND >= 30 && KIND <= 49) { // which single-rate VALU opcodes run at the fast (fp32-FMA-class) rate
            unsigned long long cm[2] = {0, 0};
            REP16(for (int i = 0; i < 8; ++i) {
                if (KIND == 30) asm volatile("v_max_i32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 31) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 32) asm volatile("v_and_b32_e32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (KIND == 33) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 34) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 35) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(a[i]));
                if (KIND == 36) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 37) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");
                if (KIND == 38) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(u[i]));
                if (KIND == 39) asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(a[i]));
                if (KIND == 40) asm volatile("v_mul_f32_e32 %0, 0x3f800123, %0" : "+v"(a[i]));
                if (KIND == 41) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 42) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (KIND == 43) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 44) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 45) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 46) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(0x5555aaaa3333ccccull));
                if (KIND == 47) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(cm[i & 1]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 48) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(u[i & 7] & 252));
                if (KIND == 49) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            })
        } else if (KIND >= 50 && KIND <= 57) { // dependent-issue distance: chains of 1, 2, 4 registers
            constexpr int D = (KIND & 3) == 0 ? 1 : (KIND & 3) == 1 ? 2 : (KIND & 3) == 2 ? 4 : 8;
            REP16(for (int i = 0; i < 8; ++i) {
                if (KIND < 54) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i % D]));
                else asm volatile("v_mul_f32_e32 %0, %0, %0" : "+v"(a[i % D]));
            })
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
    for (int wpc : {4, 8, 16, 32}) {
        if (KIND >= 11 && KIND < 50 && wpc != 8 && wpc != 32) continue;
        if (KIND >= 30 && KIND < 50 && wpc != 32) continue;
        if (KIND >= 50 && wpc == 32) continue;          // waves per CU -> waves per SIMD = wpc/4
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

int main(int argc, char **argv) {
    if (argc > 2) {      // dependent-issue distances
        run<50>("v_pk_mul_f32 dist 1", 128);
        run<51>("v_pk_mul_f32 dist 2", 128);
        run<52>("v_pk_mul_f32 dist 4", 128);
        run<53>("v_pk_mul_f32 dist 8", 128);
        run<54>("v_mul_f32 dist 1", 128);
        run<55>("v_mul_f32 dist 2", 128);
        run<56>("v_mul_f32 dist 4", 128);
        run<57>("v_mul_f32 dist 8", 128);
        return 0;
    }
    if (argc > 1) {      // transcendental / approx-EMD kinds only
        run<11>("v_exp_f32", 128);
        run<12>("emd step (9 VALU incl. exp)", 128 * 9);
        run<13>("v_sub_f32_e32 v,v", 128);
        run<14>("v_sub_f32_e32 s,v", 128);
        run<15>("v_mul_f32_e32 v,v", 128);
        run<16>("v_fmac_f32_e32 v,v", 128);
        run<17>("v_fmac_f32_e32 s,v", 128);
        run<18>("v_fma_f32 s,1.0,-v", 128);
        run<19>("v_fma_f32 v,v,v", 128);
        run<20>("v_perm_b32 v,v,v", 128);
        run<21>("v_perm_b32 v,v,s", 128);
        run<22>("v_pk_fma_f32 v,v,v", 128);
        run<23>("v_pk_fma_f32 v,s,v", 128);
        run<24>("v_pk_add_f32 v,v", 128);
        run<25>("v_pk_add_f32 s(bcast),v", 128);
        run<26>("v_pk_mul_f32 v,v", 128);
        run<27>("v_cvt_pk_bf16_f32", 128);
        run<28>("v_max_i32 0,v", 128);
        run<29>("v_min3_f32 v,v,v", 128);
        run<30>("v_max_i32 v,v", 128);
        run<31>("v_and_b32 v,v", 128);
        run<32>("v_and_b32 literal,v", 128);
        run<33>("v_add_u32 v,v", 128);
        run<34>("v_max_f32 v,v", 128);
        run<35>("v_add_f32 1.0,v", 128);
        run<36>("v_mov_b32 v,v", 128);
        run<37>("v_cndmask_b32 v,v,vcc", 128);
        run<38>("v_lshlrev_b32 1,v", 128);
        run<39>("v_max_f32 0,v", 128);
        run<40>("v_mul_f32 literal,v", 128);
        run<41>("v_sub_f32 v,v", 128);
        run<42>("v_and_or_b32 v,v,v", 128);
        run<43>("v_fma_f32 v,v,-v", 128);
        run<44>("v_add_f32_e64 v,|v|", 128);
        run<45>("v_min_f32 v,v", 128);
        run<46>("v_cndmask_b32_e64 v,v,s[2]", 128);
        run<47>("v_cmp_lt_f32_e64 s[2],v,v", 128);
        run<48>("ds_bpermute_b32 + wait", 128);
        run<49>("v_mov_b32_dpp quad_perm", 128);
        return 0;
    }
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
