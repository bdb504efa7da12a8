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
// addresses, not contents.  Disabled with DPF_TRAIN_GRAPH=0, bypassed while the stream is already being captured.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

inline std::atomic<long> &dpf_graph_replay_counter() { static std::atomic<long> n{0}; return n; }   // graph launches, process-wide

struct GraphKey {
    uint64_t h = 1469598103934665603ull;           // FNV-1a over everything that shapes the launch sequence
    void add(const void *p, size_t n) {
        const uint8_t *b = (const uint8_t *)p;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    }
    template <class T> void val(const T &v) { add(&v, sizeof(T)); }
};

class GraphCache {
    static constexpr int SLOTS = 8;
    struct Entry { uint64_t key = 0; hipGraphExec_t exec = nullptr; int state = 0; uint64_t stamp = 0; };   // state: 0 free, 1 seen once, 2 graph, 3 not capturable
    Entry e_[SLOTS];
    uint64_t clock_ = 0;
    std::mutex mu_;

public:
    static bool enabled() {
        static const bool on = !(getenv("DPF_TRAIN_GRAPH") && atoi(getenv("DPF_TRAIN_GRAPH")) == 0);
        return on;
    }
    // direct(st): issues the launches on stream st, returns 0 or an error code.  Recording happens on a private stream
    // (the caller's may be the legacy default stream, which cannot be captured); the graph is launched on the caller's.
    template <class F>
    int run(uint64_t key, hipStream_t s, F &&direct) {
        if (!enabled()) return direct(s);
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s != nullptr && (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone)) {
            (void)hipGetLastError();
            return direct(s);
        }
        std::lock_guard<std::mutex> lock(mu_);
        Entry *hit = nullptr, *victim = &e_[0];
        for (Entry &e : e_) {
            if (e.state && e.key == key) { hit = &e; break; }
            if (e.stamp < victim->stamp) victim = &e;
        }
        if (!hit) {                                    // first sighting: eager
            if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
            *victim = Entry{key, nullptr, 1, ++clock_};
            return direct(s);
        }
        hit->stamp = ++clock_;
        if (hit->state == 3) return direct(s);
        if (hit->state == 1) {                         // second sighting: record
            hipStream_t cs = nullptr;
            hipGraph_t g = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess ||
                hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                if (cs) (void)hipStreamDestroy(cs);
                (void)hipGetLastError();
                hit->state = 3;
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
                return direct(s);                      // nothing ran during the failed recording
            }
            (void)hipGraphDestroy(g);
            hit->state = 2;
        }
        dpf_graph_replay_counter().fetch_add(1, std::memory_order_relaxed);
        return (int)hipGraphLaunch(hit->exec, s);
    }
};
