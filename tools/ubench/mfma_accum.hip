// Micro-test (gfx950): how does v_mfma_f32_32x32x16_f16 add its 16 products?  Three products per output, +X, -X and a small t,
// placed in different K slots (and X / t of different sizes): an fp32 adder that rounds after every product loses t whenever it
// is added while the partial sum is +-X (t < ulp(X) / 2); an adder that keeps the products exact until one final rounding
// returns t for every placement.  What this decides: whether the expanded-form squared distance of csrc/emd.hip (terms of size
// 4^j |q'|^2 that cancel to O(1)) is limited by the accumulation or only by its operands' 22 bits.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_accum.hip -o mfma_accum ; run: ./mfma_accum
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// every row of A is a[16], every column of B is b[16]: all 1024 outputs are the same sum over k of a[k] * b[k]
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
    const int places[][3] = {{0, 1, 2}, {2, 1, 0}, {0, 2, 1}, {0, 7, 8}, {0, 8, 15}, {8, 0, 15}, {15, 8, 0}, {3, 12, 5}, {4, 5, 6}, {7, 8, 9}};
    const float Xs[] = {4096.f, 16384.f, 60000.f};
    const float ts[] = {1.0f, 0.0078125f, 0.000244140625f, 5.9604645e-8f * 64};
    printf("sum of (+X) + (t) + (-X) in K slots (kX, kt, k-X); exact answer t\n");
    for (float X : Xs)
        for (float t : ts) {
            printf("X = %-8g t = %-12g:", X, t);
            for (auto &pl : places) {
                for (int k = 0; k < 16; ++k) { a[k] = (_Float16)0.f; b[k] = (_Float16)0.f; }
                a[pl[0]] = (_Float16)X; b[pl[0]] = (_Float16)1.f;
                // t as a product of two fp16 numbers (t may be below fp16's normal range on its own)
                const float r = sqrtf(t);
                a[pl[1]] = (_Float16)r; b[pl[1]] = (_Float16)(t / (float)(_Float16)r);
                a[pl[2]] = (_Float16)(-X); b[pl[2]] = (_Float16)1.f;
                const float texact = (float)a[pl[1]] * (float)b[pl[1]];
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, out);
                hipDeviceSynchronize();
                printf(" %s", out[0] == texact ? "t" : out[0] == 0.f ? "0" : "?");
            }
            printf("\n");
        }
    return 0;
}
