// Zero-fill as a KERNEL node, never hipMemsetAsync.
//
// r03 finding (tests/diag/replay_vs_eager.py, profiles/r03_replay_vs_eager.txt): a hipMemsetAsync recorded by stream
// capture becomes a memset node that this ROCm (7.2, gfx950) does not reliably order before the kernel node that follows
// it when the graph is replayed -- the training backward's tickets were now and then cleared AFTER the first pass-1
// workgroups had arrived, a cloud lost its "last arriver", and the step used the previous step's per-cloud totals
// (losses drifted from the eager path's from step ~22 on).  A fill kernel in the same position is ordered like every
// other kernel node and costs the same launch.  Every entry point of this library may be called under capture (ours or
// the caller's torch.cuda.graph), so none of them issues a memset.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ __launch_bounds__(256) void dpf_zero_words_kernel(uint32_t *__restrict__ p, size_t nwords) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) p[i] = 0u;
}
}  // namespace

// bytes must be a multiple of 4 and p 4-byte aligned (every caller clears float / uint32 arrays)
static inline hipError_t dpf_zero_async(void *p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return hipErrorInvalidValue;
    const size_t nwords = bytes / 4;
    hipLaunchKernelGGL(dpf_zero_words_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, s, (uint32_t *)p, nwords);
    return hipGetLastError();
}
