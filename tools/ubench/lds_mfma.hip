// Micro-benchmark: how fast can the four SIMDs of a CU stream 1 KiB MFMA operand fragments out of LDS
// (ds_read_b128, lane l reads 16 B at base + 16 l), alone and feeding one or two MFMAs per fragment?
// Build: hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    u32x4 x = {0, 0, 0, 0};
    const u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 16; f += 2) {
            const u32x4 a0 = *(const u32x4 *)(smem + ((it * 16 + f) & 63) * 1024 + lane * 16);
            const u32x4 a1 = *(const u32x4 *)(smem + ((it * 16 + f + 1) & 63) * 1024 + lane * 16);
            if (KIND == 0) {            // reads only
                x ^= a0; x ^= a1;
            } else if (KIND == 1) {     // one MFMA per fragment
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
            } else {                    // two MFMAs per fragment
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b), acc2, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b), acc3, 0, 0, 0);
            }
        }
    }
    float s = (float)(x.x ^ x.y ^ x.z ^ x.w);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int mfma_per_frag, int waves) {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 65536, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double frags_per_cu = (double)waves * iters * 16;       // waves x 16 fragments per iteration
    const double ns_per_frag = ms * 1e6 / frags_per_cu;
    printf("%-22s %2d waves/CU %.3f ms: %.2f ns per 1 KiB fragment per CU (%.0f B/ns/CU); per SIMD: one fragment every %.1f ns, %d MFMA each\n", name, waves, ms,
           ns_per_frag, 1024.0 / ns_per_frag, ns_per_frag * 4, mfma_per_frag);
    hipFree(d);
}
int main() {
    for (int w : {4, 8, 16}) {
        run<0>("ds_read_b128 only", 0, w);
        run<1>("1 MFMA per fragment", 1, w);
        run<2>("2 MFMA per fragment", 2, w);
    }
    return 0;
}
