// Chip peaks confirmed on the box ("builder must confirm on the box"): HBM stream copy (float4, 1 GiB in + 1 GiB
// out), dense bf16 MFMA rate (v_mfma_f32_32x32x16_bf16, 4 accumulators per wave, 2 waves per SIMD) and fp32 VALU FMA
// rate (v_pk_fma_f32).  build: hipcc --offload-arch=gfx950 -O3 peaks.hip -o peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(512) void mfma_kernel(float *out, int iters) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(1024) void fma_kernel(float *out, int iters, float s) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{s + i + threadIdx.x, s - i};
    const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], m, c);
    float t = 0;
    for (int i = 0; i < 8; ++i) t += x[i].x + x[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = t;
}
static float time_ms(void (*launch)(void *), void *p, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(p);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(p);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
struct CopyArgs { float4 *in, *out; size_t n; };
struct KArgs { float *out; int iters; };
int main() {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    printf("device: %s  CUs %d  clock %d MHz  memory clock %d MHz  bus %d bit  L2 %d KiB  HBM %.0f GiB\n", pr.name, pr.multiProcessorCount,
           pr.clockRate / 1000, pr.memoryClockRate / 1000, pr.memoryBusWidth, pr.l2CacheSize / 1024, pr.totalGlobalMem / 1073741824.0);
    CopyArgs ca; ca.n = (size_t)1 << 26;   // 64 Mi float4 = 1 GiB
    (void)hipMalloc(&ca.in, ca.n * 16); (void)hipMalloc(&ca.out, ca.n * 16);
    (void)hipMemset(ca.in, 1, ca.n * 16);
    float ms = time_ms([](void *p) { CopyArgs *a = (CopyArgs *)p; hipLaunchKernelGGL(copy_kernel, dim3(256 * 32), dim3(256), 0, 0, a->in, a->out, a->n); }, &ca, 10);
    printf("HBM stream copy (float4, 1 GiB read + 1 GiB write): %.3f ms = %.0f GB/s  (spec 8000; guide: 6290 measured)\n", ms, 2.0 * ca.n * 16 / ms / 1e6);
    KArgs ka; ka.iters = 4000;
    (void)hipMalloc(&ka.out, 256 * 8 * 1024 * sizeof(float));
    ms = time_ms([](void *p) { KArgs *a = (KArgs *)p; hipLaunchKernelGGL(mfma_kernel, dim3(256 * 2), dim3(512), 0, 0, a->out, a->iters); }, &ka, 5);
    printf("bf16 MFMA 32x32x16 (512 WGs x 8 waves): %.3f ms = %.0f TFLOP/s  (dense spec 2500)\n", ms,
           512.0 * 8 * ka.iters * 4 * 2.0 * 32 * 32 * 16 / ms / 1e9);
    ms = time_ms([](void *p) { KArgs *a = (KArgs *)p; hipLaunchKernelGGL(fma_kernel, dim3(256 * 2), dim3(1024), 0, 0, a->out, a->iters, 1.0f); }, &ka, 5);
    printf("fp32 VALU v_pk_fma_f32 (512 WGs x 16 waves): %.3f ms = %.1f TFLOP/s  (spec 157.3)\n", ms,
           512.0 * 1024 * ka.iters * 8 * 4.0 / ms / 1e9);
    return 0;
}
