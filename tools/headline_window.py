"""Why did the driver's 20-step window read 85 us / step when the default-length run reads 70-73?

One process, the workload of `bench.py` (configs[1]), the timed region bracketed several ways, each repeated R times after the
same settle.  Prints one JSON object: per bracketing the min / median / max us per step.

    python tools/headline_window.py [--reps 30] [--steps 20] [--warmup 5]

  wall_singles_then_multi   r05's region: W single-step replays, synchronize, perf_counter around K/G multi-step replays
  wall_multi_then_multi     the same with the warm-up rounded up to whole multi-step replays
  event_*                   the same regions timed by HIP events recorded on the launch stream
  event_preroll             one more multi replay in flight in front of the start event (GPU never idle before the window)
  wall_idle_ms_X            r05's region after the host slept X ms behind the synchronize (what an idle gap costs)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=400)
    a = ap.parse_args()
    args = bench.parse(["--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup)])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, device, 32)
    L = args.layers
    step = bench.make_step(dec, z, g, tgt_pm, L)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        step()
    torch.cuda.synchronize()
    G = args.graph_steps
    single = torch.cuda.CUDAGraph()
    with torch.cuda.graph(single, stream=st):
        step()
    multi = torch.cuda.CUDAGraph()
    with torch.cuda.graph(multi, stream=st):
        for _ in range(G):
            step()
    K, W = a.steps, a.warmup

    def run(n, singles_for_remainder=True):
        with torch.cuda.stream(st):
            for _ in range(n // G):
                multi.replay()
            for _ in range(n % G):
                single.replay()

    def region(kind, idle_ms=0.0):
        run(a.settle)
        if kind == "singles":
            run(W)
        else:
            run(((W + G - 1) // G) * G)
        torch.cuda.synchronize()
        if idle_ms:
            time.sleep(idle_ms * 1e-3)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            if kind == "preroll":
                multi.replay()
            t0 = time.perf_counter()
            s.record()
        run(K)
        with torch.cuda.stream(st):
            e.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e6
        return wall, s.elapsed_time(e) / K * 1e3

    out = {"steps": K, "warmup": W, "graph_steps": G, "reps": a.reps, "sclk_mhz": bench.device_clock_mhz(),
           "device": torch.cuda.get_device_name(0)}

    def stats(v):
        v = np.asarray(v)
        return {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max()), "p90": float(np.percentile(v, 90))}

    for kind in ("singles", "multi", "preroll"):
        w, ev = zip(*[region(kind) for _ in range(a.reps)])
        out["wall_%s" % kind] = stats(w)
        out["event_%s" % kind] = stats(ev)
    for idle in (1.0, 10.0, 100.0, 1000.0):
        w, ev = zip(*[region("singles", idle) for _ in range(max(3, a.reps // 5))])
        out["wall_idle_ms_%g" % idle] = stats(w)
        out["event_idle_ms_%g" % idle] = stats(ev)
    # the default-length window for reference
    run(a.settle)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(1000)
    torch.cuda.synchronize()
    out["wall_1000_steps"] = (time.perf_counter() - t0) / 1000 * 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
