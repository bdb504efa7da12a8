// What does a grid-wide rendezvous cost INSIDE a persistent kernel on gfx950 (256 workgroups, one per CU, 8 XCDs with
// non-coherent L2s), against the ~4 us a dependent kernel launch costs?  (r03: the training step is 6 dependent launches per
// layer, each with ~3.5-4 us of fixed cost; a persistent kernel would replace launches by barriers.)
//   variant 0: barrier only -- one relaxed agent-scope fetch_add per workgroup on a monotonic counter, then polling loads
//   variant 1: + data: every workgroup publishes a row of 512 floats (write-through stores, s_waitcnt vmcnt(0)) before the
//              barrier and sums 32 rows (agent-scope loads, 8 in flight) after it -- the shape of the per-cloud totals
//   variant 2: + exact accumulation: every workgroup adds 512 values into SHARED 3-limb fixed-point accumulators with
//              64-bit integer atomics (order-independent, hence deterministic) before the barrier, every workgroup reads
//              the 512 totals after it
// The spin is BOUNDED (a workgroup that is not co-resident can never arrive): on time-out a sticky flag ends all waiting.
// build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier ; run: ./grid_barrier [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct Bar {
    unsigned *count;       // monotonic arrival counter
    unsigned *fail;        // sticky time-out flag
};

__device__ __forceinline__ bool grid_barrier(const Bar &b, unsigned target, unsigned *sh) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's published data has left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        int spins = 0;
        while (__hip_atomic_load(b.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 20) || __hip_atomic_load(b.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(b.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *sh = ok;
    }
    __syncthreads();
    return *sh != 0;
}

// 3-limb fixed point: value * 2^40 as a 126-bit integer in limbs of 42 bits (each limb has 22 bits of head room)
__device__ __forceinline__ void limbs_of(double v, long long out[3]) {
    const double s = v * 1099511627776.0;                   // 2^40
    const double hi = trunc(s * (1.0 / 4398046511104.0));   // / 2^42
    const double r0 = s - hi * 4398046511104.0;
    const double top = trunc(hi * (1.0 / 4398046511104.0));
    const double r1 = hi - top * 4398046511104.0;
    out[0] = (long long)r0; out[1] = (long long)r1; out[2] = (long long)top;
}

template <int VAR>
__global__ __launch_bounds__(512) void persist_kernel(Bar b, int rounds, float *rows, unsigned long long *acc, float *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    unsigned *sh = (unsigned *)smem;
    const int nwg = gridDim.x, wg = blockIdx.x;
    float keep = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (VAR == 1) {
            float *row = rows + ((size_t)(r & 1) * nwg + wg) * 512;
            __hip_atomic_store(&row[threadIdx.x], (float)(wg + r) + threadIdx.x * 0.001f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (VAR == 2) {
            long long l[3];
            limbs_of((double)((wg % 7) - 3) * 0.37 + threadIdx.x * 1e-3, l);
            unsigned long long *a = acc + ((size_t)(r & 1) * 512 + threadIdx.x) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) __hip_atomic_fetch_add(&a[i], (unsigned long long)l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier(b, (unsigned)(r + 1) * nwg, sh)) break;
        if (VAR == 1) {
            const int c0 = (wg / 8) * 8;                    // "my cloud": 8 workgroups ... read 32 rows to mimic B = 32 totals
            float s = 0.f;
            for (int k0 = 0; k0 < 32; k0 += 8) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = __hip_atomic_load(&rows[((size_t)(r & 1) * nwg + (c0 + k0 + i) % nwg) * 512 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
            }
            keep += s;
        }
        if (VAR == 2) {
            const unsigned long long *a = acc + ((size_t)(r & 1) * 512 + threadIdx.x) * 3;
            long long l[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) l[i] = (long long)__hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double t = (((double)l[2] * 4398046511104.0 + (double)l[1]) * 4398046511104.0 + (double)l[0]) * (1.0 / 1099511627776.0);
            keep += (float)t;
        }
    }
    out[(size_t)wg * 512 + threadIdx.x] = keep;
}

__global__ __launch_bounds__(512) void tiny_kernel(float *out) { if (threadIdx.x == 0) out[blockIdx.x] += 1.f; }

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const int nwg = 256, lds = 140 * 1024;
    unsigned *ctr; float *rows, *out; unsigned long long *acc;
    (void)hipMalloc(&ctr, 256); (void)hipMalloc(&rows, (size_t)2 * nwg * 512 * 4); (void)hipMalloc(&out, (size_t)nwg * 512 * 4);
    (void)hipMalloc(&acc, (size_t)2 * 512 * 3 * 8);
    (void)hipFuncSetAttribute((const void *)persist_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void *)persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int var = 0; var < 3; ++var)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(ctr, 0, 256); (void)hipMemset(acc, 0, (size_t)2 * 512 * 3 * 8);
            Bar b = {ctr, ctr + 16};
            (void)hipDeviceSynchronize();
            float ms1 = 0, ms2 = 0;
            for (int pass = 0; pass < 2; ++pass) {          // rounds and 1 round: the difference is the per-round cost
                const int R = pass == 0 ? rounds : 1;
                (void)hipMemset(ctr, 0, 256);
                (void)hipEventRecord(e0);
                if (var == 0) hipLaunchKernelGGL(persist_kernel<0>, dim3(nwg), dim3(512), lds, 0, b, R, rows, acc, out);
                if (var == 1) hipLaunchKernelGGL(persist_kernel<1>, dim3(nwg), dim3(512), lds, 0, b, R, rows, acc, out);
                if (var == 2) hipLaunchKernelGGL(persist_kernel<2>, dim3(nwg), dim3(512), lds, 0, b, R, rows, acc, out);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(pass == 0 ? &ms1 : &ms2, e0, e1);
            }
            unsigned fail = 0;
            (void)hipMemcpy(&fail, ctr + 16, 4, hipMemcpyDeviceToHost);
            printf("variant %d: %d rounds %.1f us, 1 round %.1f us -> %.3f us per round%s\n", var, rounds, ms1 * 1e3, ms2 * 1e3,
                   (ms1 - ms2) * 1e3 / (rounds - 1), fail ? "   ** TIMED OUT **" : "");
        }
    // reference: the same number of dependent tiny launches (stream order)
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(tiny_kernel, dim3(nwg), dim3(512), 0, 0, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("dependent tiny launches (eager, stream order): %.3f us each\n", ms * 1e3 / rounds);
    }
    return 0;
}
