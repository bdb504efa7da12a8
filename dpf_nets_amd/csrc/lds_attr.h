// Raising a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE
// property of the function: a process that launches on cuda:0 and then on cuda:1 must set it on both.
// One LdsLimit per kernel instantiation remembers, per device ordinal, the largest size already granted;
// concurrent callers may both set the attribute (idempotent), never skip it.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

struct LdsLimit {
    static constexpr int MAX_DEV = 64;
    std::atomic<int> granted[MAX_DEV];
    hipError_t ensure(const void *fn, int bytes) {
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool tracked = dev >= 0 && dev < MAX_DEV;
        if (tracked && granted[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        if (tracked) {
            int cur = granted[dev].load(std::memory_order_relaxed);
            while (cur < bytes && !granted[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        }
        return hipSuccess;
    }
};
