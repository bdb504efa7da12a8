// Launch-bound call sequences as hipGraphs.
//
// A training-mode stack call issues ~6 (forward) / ~5 (backward) small dependent kernels per layer -- 700 launches
// per step at 63 layers, ~9 us of host time each: once the kernels were tuned the step was bound by the CPU's launch
// rate, not by the GPU.  GraphCache turns a call into ONE hipGraphLaunch: the first time a key (every scalar and every
// pointer the call would pass to its kernels) is seen the call runs eagerly (which also performs the per-device
// one-time setup such as raising dynamic-LDS limits), the second time it is recorded with stream capture and
// instantiated, from then on it is replayed.  A training loop presents the same pointers step after step (PyTorch's
// caching allocator hands the same blocks back), so the steady state is all replays; any change of shape, mode,
// precision or address is a different key and takes the eager path first.  Data may change freely: the graph holds
// addresses, not contents.  Disabled with DPF_TRAIN_GRAPH=0 or dpf_train_graph_set_enabled(0), bypassed while the
// stream is already being captured.
//
// A hit is verified: the entry keeps the key BYTES (a few hundred) and a replay needs hash AND bytes to match -- a
// 64-bit collision would otherwise replay a graph holding another call's addresses.  Counters (dpf_train_graph_stats)
// let a training loop detect thrash: more than SLOTS live pointer sets, or an allocator that does not settle, shows up
// as `evictions` growing and `replays` standing still.
//
// Only KERNEL nodes are recorded: every zero-fill of the recorded sequences is a kernel (zero_fill.h) because a captured
// hipMemsetAsync node was found not to be ordered reliably before its successor on replay (ROCm 7.2, gfx950).
//
// Capture runs on a private non-blocking stream in ThreadLocal mode inside the library call.  A caller that is itself
// capturing on the SAME thread is detected (the call passes through into the caller's capture); a caller capturing on
// ANOTHER thread in Global mode is not visible from here -- do not record a torch.cuda.graph on one thread while a
// second thread makes its first two training calls ("graphs").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

struct GraphStats {
    std::atomic<long> replays{0}, eager{0}, records{0}, evictions{0}, uncapturable{0};
};
inline GraphStats &dpf_graph_stats() { static GraphStats s; return s; }                 // process-wide
inline std::atomic<int> &dpf_graph_enabled_flag() {
    static std::atomic<int> on{!(getenv("DPF_TRAIN_GRAPH") && atoi(getenv("DPF_TRAIN_GRAPH")) == 0)};
    return on;
}

struct GraphKey {
    uint64_t h = 1469598103934665603ull;           // FNV-1a over everything that shapes the launch sequence
    std::vector<uint8_t> bytes;                    // ... and the bytes themselves, compared on a hit
    void add(const void *p, size_t n) {
        const uint8_t *b = (const uint8_t *)p;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
        bytes.insert(bytes.end(), b, b + n);
    }
    template <class T> void val(const T &v) { add(&v, sizeof(T)); }
};

class GraphCache {
    static constexpr int SLOTS = 8;
    struct Entry {
        uint64_t hash = 0;
        std::vector<uint8_t> key;
        hipGraphExec_t exec = nullptr;
        int state = 0;                             // 0 free, 1 seen once, 2 graph, 3 not capturable
        uint64_t stamp = 0;
    };
    Entry e_[SLOTS];
    uint64_t clock_ = 0;
    std::mutex mu_;

public:
    static bool enabled() { return dpf_graph_enabled_flag().load(std::memory_order_relaxed) != 0; }
    // direct(st): issues the launches on stream st, returns 0 or an error code.  Recording happens on a private stream
    // (the caller's may be the legacy default stream, which cannot be captured); the graph is launched on the caller's.
    template <class F>
    int run(const GraphKey &key, hipStream_t s, F &&direct) {
        GraphStats &gs = dpf_graph_stats();
        if (!enabled()) return direct(s);
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s != nullptr && (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone)) {
            (void)hipGetLastError();
            return direct(s);
        }
        std::lock_guard<std::mutex> lock(mu_);
        Entry *hit = nullptr, *victim = &e_[0];
        for (Entry &e : e_) {
            if (e.state && e.hash == key.h && e.key == key.bytes) { hit = &e; break; }
            if (e.stamp < victim->stamp) victim = &e;
        }
        if (!hit) {                                    // first sighting: eager
            if (victim->state) gs.evictions.fetch_add(1, std::memory_order_relaxed);
            if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
            victim->hash = key.h; victim->key = key.bytes; victim->exec = nullptr; victim->state = 1; victim->stamp = ++clock_;
            gs.eager.fetch_add(1, std::memory_order_relaxed);
            return direct(s);
        }
        hit->stamp = ++clock_;
        if (hit->state == 3) { gs.eager.fetch_add(1, std::memory_order_relaxed); return direct(s); }
        if (hit->state == 1) {                         // second sighting: record
            hipStream_t cs = nullptr;
            hipGraph_t g = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess ||
                hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                if (cs) (void)hipStreamDestroy(cs);
                (void)hipGetLastError();
                hit->state = 3;
                gs.uncapturable.fetch_add(1, std::memory_order_relaxed);
                gs.eager.fetch_add(1, std::memory_order_relaxed);
                return direct(s);
            }
            const int rc = direct(cs);
            const hipError_t ec = hipStreamEndCapture(cs, &g);
            (void)hipStreamDestroy(cs);
            if (rc != 0 || ec != hipSuccess || g == nullptr ||
                hipGraphInstantiate(&hit->exec, g, nullptr, nullptr, 0) != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                hit->exec = nullptr;
                hit->state = 3;
                gs.uncapturable.fetch_add(1, std::memory_order_relaxed);
                gs.eager.fetch_add(1, std::memory_order_relaxed);
                return direct(s);                      // nothing ran during the failed recording
            }
            (void)hipGraphDestroy(g);
            hit->state = 2;
            gs.records.fetch_add(1, std::memory_order_relaxed);
        }
        gs.replays.fetch_add(1, std::memory_order_relaxed);
        return (int)hipGraphLaunch(hit->exec, s);
    }
};
