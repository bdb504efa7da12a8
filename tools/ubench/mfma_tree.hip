// Micro-test (gfx950): in which ORDER does v_mfma_f32_32x32x16_f16 add its 16 products?  For every pair of K slots (i, j): +X in
// i, -X in j, a small t in every other slot k in turn -- if the result is t for EVERY k, slots i and j are added to each other
// before anything else joins them.  Then triples: which third slot k may hold a term that is added right after the pair.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_tree.hip -o mfma_tree ; run: ./mfma_tree
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const _Float16 *a, const _Float16 *b, float *out) {
    const int lane = threadIdx.x, half = lane >> 5;
    h8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[half * 8 + i]; bv[i] = b[half * 8 + i]; }
    f16v acc;
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 2" : "=&v"(acc) : "v"(av), "v"(bv));
    if (lane == 0) out[0] = acc[0];
}

int main() {
    _Float16 *a, *b;
    float *out;
    hipMallocManaged(&a, 32); hipMallocManaged(&b, 32); hipMallocManaged(&out, 4);
    const float X = 16384.f, t = 0.000244140625f;      // t = 2^-12: lost whenever it is added to a partial sum of size X
    auto run = [&](int i, int j, int k) {
        for (int q = 0; q < 16; ++q) { a[q] = (_Float16)0.f; b[q] = (_Float16)0.f; }
        a[i] = (_Float16)X; b[i] = (_Float16)1.f;
        a[j] = (_Float16)(-X); b[j] = (_Float16)1.f;
        a[k] = (_Float16)0.015625f; b[k] = (_Float16)0.015625f;   // 2^-6 * 2^-6 = t
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, out);
        hipDeviceSynchronize();
        return out[0] == t;
    };
    printf("rows i, columns j: number of third slots k (of 14) for which (+X at i) + (t at k) + (-X at j) returns t exactly\n    ");
    for (int j = 0; j < 16; ++j) printf("%3d", j);
    printf("\n");
    for (int i = 0; i < 16; ++i) {
        printf("%3d:", i);
        for (int j = 0; j < 16; ++j) {
            if (i == j) { printf("  ."); continue; }
            int ok = 0;
            for (int k = 0; k < 16; ++k) if (k != i && k != j) ok += run(i, j, k);
            printf("%3d", ok);
        }
        printf("\n");
    }
    printf("for (i, j) = (0, 1) and (0, 8) and (3, 5): which k keep t: ");
    const int pr[3][2] = {{0, 1}, {0, 8}, {3, 5}};
    for (auto &p : pr) {
        printf(" (%d,%d):", p[0], p[1]);
        for (int k = 0; k < 16; ++k) if (k != p[0] && k != p[1]) printf("%s", run(p[0], p[1], k) ? " +" : " -"), printf("%d", k);
    }
    printf("\n");
    return 0;
}
