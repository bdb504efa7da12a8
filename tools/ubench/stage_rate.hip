// How fast can EVERY CU of the chip pull the same `KB` of packed weights into its LDS at kernel start?  (r03: the
// training kernels stage 60-92 KB before their first MFMA; 256 workgroups ask the L2 for the same lines at the same time.)
//   variant 0: global_load_lds, pieces in the same order on every workgroup (what stage_bytes did)
//   variant 1: global_load_lds, piece order rotated by the workgroup index (different CUs hit different L2 channels)
//   variant 2: global_load_dwordx4 into VGPRs (all of a wave's pieces in flight), then ds_write_b128; same order everywhere
//   variant 3: variant 2 with the rotation
// Each workgroup (512 threads) stages, barriers, and folds a checksum of its LDS so nothing is optimised away; the kernel is
// timed with HIP events over `reps` launches; kernel duration minus the empty-kernel duration ~ the staging time.
// build: hipcc --offload-arch=gfx950 -O3 stage_rate.hip -o stage_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(512) void stage_kernel(const uint8_t *__restrict__ src, int npieces, uint32_t *__restrict__ out, int rotmul) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rot = (VAR & 1) ? (int)((blockIdx.x * (unsigned)rotmul) % (unsigned)npieces) : 0;
    if (VAR < 2) {
        for (int c = wave; c < npieces; c += 8) {
            int p = c + rot; if (p >= npieces) p -= npieces;
            __builtin_amdgcn_global_load_lds((glb_void *)(src + p * 1024 + lane * 16), (lds_void *)(smem + p * 1024), 16, 0, 0);
        }
    } else {
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = wave + 8 * i;
            int p = c + rot; if (p >= npieces) p -= npieces;
            if (c < npieces) v[i] = *(const u32x4 *)(src + p * 1024 + lane * 16);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = wave + 8 * i;
            int p = c + rot; if (p >= npieces) p -= npieces;
            if (c < npieces) *(u32x4 *)(smem + p * 1024 + lane * 16) = v[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < npieces * 256; i += 512) s += ((const uint32_t *)smem)[i];
    if (s == 0x12345678u) out[blockIdx.x] = s;          // (never true for the test pattern: keeps the reads alive)
    if (threadIdx.x == 0) out[blockIdx.x] = ((const uint32_t *)smem)[(npieces - 1) * 256 + 255];
}
__global__ __launch_bounds__(512) void empty_kernel(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = 1; }

static float time_us(int var, const uint8_t *src, int npieces, uint32_t *out, int nwg, int reps, int rotmul) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&]() {
        const int lds = npieces * 1024;
        switch (var) {
        case -1: hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(512), 0, 0, out); break;
        case 0: hipLaunchKernelGGL(stage_kernel<0>, dim3(nwg), dim3(512), lds, 0, src, npieces, out, rotmul); break;
        case 1: hipLaunchKernelGGL(stage_kernel<1>, dim3(nwg), dim3(512), lds, 0, src, npieces, out, rotmul); break;
        case 2: hipLaunchKernelGGL(stage_kernel<2>, dim3(nwg), dim3(512), lds, 0, src, npieces, out, rotmul); break;
        default: hipLaunchKernelGGL(stage_kernel<3>, dim3(nwg), dim3(512), lds, 0, src, npieces, out, rotmul); break;
        }
    };
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    uint8_t *src; uint32_t *out;
    const int maxkb = 128;
    (void)hipMalloc(&src, maxkb * 1024); (void)hipMalloc(&out, 4096 * 4);
    uint32_t *h = (uint32_t *)malloc(maxkb * 1024);
    for (int i = 0; i < maxkb * 256; ++i) h[i] = 2654435761u * (i + 1);
    (void)hipMemcpy(src, h, maxkb * 1024, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)stage_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)stage_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)stage_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nwg = 256, reps = 200;
    const float t_empty = time_us(-1, src, 0, out, nwg, reps, 0);
    printf("empty kernel (256 workgroups x 512 threads, back-to-back launches): %.2f us\n", t_empty);
    printf("%6s | %28s | %28s | %28s | %28s\n", "KB", "lds-dma same order", "lds-dma rotated", "vgpr same order", "vgpr rotated");
    for (int kb : {36, 60, 92, 128}) {
        printf("%6d |", kb);
        for (int var = 0; var < 4; ++var) {
            const float t = time_us(var, src, kb, out, nwg, reps, 5);
            printf(" %7.2f us (+%5.2f) %5.0f GB/s/CU |", t, t - t_empty, kb * 1024 / ((t - t_empty) * 1e-6) / 1e9);
        }
        printf("\n");
    }
    printf("rotation multiplier sweep at 92 KB (lds-dma rotated / vgpr rotated):\n");
    for (int rm : {1, 3, 5, 7, 11, 13, 23, 37}) printf("  x%-3d %7.2f %7.2f\n", rm, time_us(1, src, 92, out, nwg, reps, rm), time_us(3, src, 92, out, nwg, reps, rm));
    return 0;
}
