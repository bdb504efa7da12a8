#!/usr/bin/env python3
"""Headline benchmark: points/s through the L-layer per-point flow + Chamfer.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch resident in HBM:
  FiLM conditioner (all layers) -> fused L-layer coupling stack, mode 'direct',
  eval-BN (lib/networks/decoders.py:54-72 as evaluate() runs it, evaluating.py:70)
  -> nn_distance(out^T, target) both directions (evaluating.py:110-111)
  -> per-cloud CD reduction (evaluating.py:112).
Workload: BASELINE.json configs[1]: B=32 clouds x N=2048 points per GPU, L=14
coupling layers (first 14 layers of LocalCondRNVPDecoder(n_flows=5)), F=64, G=128.
N GPUs: one process per GPU (torchrun), every rank runs its own B=32 shard, no
data-path collective ("weak" scaling); value = all ranks' points / max-rank time.

Prints ONE JSON line (rank 0) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak
FLOP_PER_POINT_LAYER = 17152   # 2 branches x (2F|K| + 2F^2 + 2|W|F)
MFMA_PRODUCTS = {"bf16": 1, "bf16x3": 3, "bf16x6": 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip needs a few hundred steps (tens of ms) under load to settle its clocks -- with 20 warm-up steps the
    # same step measures 98-99 us, after 300+ it measures 93 us and stays there
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=14)
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--precision", default=os.environ.get("DPF_PRECISION", "bf16x3"),
                    choices=sorted(MFMA_PRODUCTS))
    ap.add_argument("--lists", action="store_true", help="also materialise the 3 x L per-layer lists")
    ap.add_argument("--settle", type=int, default=400,
                    help="untimed steps run BEFORE the --warmup steps so that the chip's clocks have settled under this load "
                         "whatever --warmup is (see above); 0 = none")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--streams", type=int, default=1,
                    help="consecutive steps go round-robin over this many streams (each with its own buffers), so the Chamfer "
                         "kernels of one step share the chip with the flow kernel of the next; 1 (default) = strictly one after "
                         "another, the regime the roofline numbers and the committed profiles describe")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="with --streams 1: also report (outside the timed region, as `pipelined`) the throughput with this many "
                         "steps in flight; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=6)
    return ap.parse_args()


def build_workload(args, device):
    from dpf_nets_amd import synthetic as FO
    from dpf_nets_amd.networks import LocalCondRNVPDecoder
    n_flows = (args.layers + 2) // 3
    state = FO.make_decoder_state(0, n_flows, 64, args.latent)
    dec = LocalCondRNVPDecoder(n_flows, 64, args.latent)
    dec.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}, strict=True)
    dec = dec.to(device).eval()
    dec.precision = args.precision
    dec.materialize_lists = bool(args.lists)
    tgt, z, g = FO.synthetic_inputs(0, args.batch, args.points, args.latent)
    z = torch.from_numpy(z).to(device)
    g = torch.from_numpy(g).to(device)
    tgt_pm = torch.from_numpy(np.ascontiguousarray(tgt.transpose(0, 2, 1))).to(device)   # (B,N,3), resident
    return dec, state, n_flows, z, g, tgt, tgt_pm


def make_step(dec, z, g, tgt_pm, L):
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    from dpf_nets_amd.networks.utils import chamfer_per_cloud
    stack = dec.stack()

    def step():
        p_out, sum_lv, ps, mus, lvs = stack.run(z, g, "direct", dec.precision, want_lists=dec.materialize_lists,
                                                n_layers=L, want_pointmajor=True)
        d1, i1, d2, i2 = BK.NNDistance(stack.last_pointmajor, tgt_pm)
        cd = chamfer_per_cloud(d1, d2)
        return p_out, d1, i1, d2, i2, cd
    return step


def time_kernel(fn, reps=20, rounds=5):
    """Average duration of one launch: HIP events (recorded on the launch stream) around a captured graph of
    `reps` back-to-back launches, so that host launch gaps do not count -- this is the number the rocprofv3
    kernel trace reports as the kernel's average duration."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    for _ in range(25):                  # settle the clocks under this kernel's load (see --warmup), as in the timed loop
        graph.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        graph.replay()
        e.record()
        e.synchronize()
        best.append(s.elapsed_time(e) / reps * 1e3)     # us
    return float(np.median(best))


def make_kernels(dec, z, g, tgt_pm, L, precision):
    """The three launches of a step as closures over their own buffers: (film, flow, nn)."""
    from dpf_nets_amd._lib import lib, PREC, MODE, current_stream
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    stack = dec.stack()
    canon, meta, packed, G = stack._ensure(precision, z.device, L)
    B, _, N = z.shape
    film = torch.empty(lib().dpf_flow_film_floats(L, B), dtype=torch.float32, device=z.device)
    p_out, sum_lv = torch.empty_like(z), torch.empty_like(z)
    pm = torch.empty((B, N, 3), dtype=torch.float32, device=z.device)
    eps = float(stack.layers[0].eps_value)

    def k_film():
        lib().dpf_flow_film(L, B, G, PREC[precision], packed.data_ptr(), g.data_ptr(), film.data_ptr(), eps, current_stream())

    def k_flow():
        lib().dpf_flow_forward(L, B, N, MODE["direct"], PREC[precision], packed.data_ptr(), meta.data_ptr(),
                               film.data_ptr(), z.data_ptr(), p_out.data_ptr(), pm.data_ptr(), sum_lv.data_ptr(),
                               None, None, None, eps, current_stream())
    d1 = torch.empty((B, N), dtype=torch.float32, device=z.device); d2 = torch.empty_like(d1)
    i1 = torch.empty((B, N), dtype=torch.int32, device=z.device); i2 = torch.empty_like(i1)

    impl = BK.NN_IMPL
    sized, fn = BK.nn_impl_entry(impl) if impl not in ("brute", "auto") else (None, None)
    nws = sized(B, N, N) if sized else 0
    ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=z.device)

    def k_nn():
        if impl in ("brute", "auto"):
            (lib().dpf_nndistance if impl == "brute" else lib().dpf_nndistance_auto)(
                B, N, pm.data_ptr(), N, tgt_pm.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(),
                current_stream())
        else:
            fn(B, N, pm.data_ptr(), N, tgt_pm.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(),
               ws.data_ptr(), nws, current_stream())
    return k_film, k_flow, k_nn


def kernel_timings(dec, z, g, tgt_pm, L, precision):
    """per-kernel average launch durations (us) measured with HIP events, each kernel alone on the chip"""
    k_film, k_flow, k_nn = make_kernels(dec, z, g, tgt_pm, L, precision)
    k_film(); k_flow(); k_nn()
    torch.cuda.synchronize()
    return {"film_kernel": time_kernel(k_film), "flow_kernel": time_kernel(k_flow), "nn_kernel": time_kernel(k_nn)}


def kernel_timings_in_flight(dec, z, g, tgt_pm, L, precision, n_streams, steps=120, warm=30):
    """The same durations in the regime of the timed loop: steps round-robin over `n_streams` streams, HIP events on the
    launch stream around each kernel -- what the rocprofv3 kernel trace of this command reports as average durations
    (kernels of neighbouring steps share the chip, so each takes longer than alone while the steps take less)."""
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    sets = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ks = make_kernels(dec, z, g, tgt_pm, L, precision)
            for k in ks:
                k()
        sets.append(ks)
    torch.cuda.synchronize()
    marks = []
    for i in range(steps):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(); sets[k][0](); ev[1].record(); sets[k][1](); ev[2].record(); sets[k][2](); ev[3].record()
        marks.append(ev)
    torch.cuda.synchronize()
    marks = marks[warm:]
    out = {}
    for j, name in enumerate(("film_kernel", "flow_kernel", "nn_kernel")):
        out[name] = float(np.mean([m[j].elapsed_time(m[j + 1]) for m in marks])) * 1e3
    return out


def cpu_baseline(args, state, n_flows, tgt, budget_s=15.0):
    """The CPU oracle (kind 'port') timed on this host's cores on a BOUNDED sample of the same
    workload: whole clouds of the B x N batch, as many as fit ~budget_s of CPU work."""
    from oracle import flow_oracle as FO
    from oracle import structural as S
    ncores = min(os.cpu_count() or 1, 32)       # torch-CPU on hundreds of threads thrashes on these small ops
    torch.set_num_threads(ncores)
    st = FO.to_torch(state)
    _, z, g = FO.synthetic_inputs(0, args.batch, args.points, args.latent)
    tgt_pm = np.ascontiguousarray(tgt.transpose(0, 2, 1))
    S.lib()

    def run(nb):
        tz, tg = torch.from_numpy(z[:nb]), torch.from_numpy(g[:nb])
        with torch.no_grad():
            ps, _, _ = FO.decoder(st, n_flows, tz, tg, "direct", n_layers=args.layers)
        out = np.ascontiguousarray(ps[-1].numpy().transpose(0, 2, 1))
        d1, _, d2, _ = S.nndistance(out, tgt_pm[:nb])
        return d1.mean(1) + d2.mean(1)
    run(1)                                       # warm-up
    t0 = time.perf_counter(); run(2); t2 = time.perf_counter() - t0          # calibration on 2 clouds
    nb = int(max(2, min(args.batch, budget_s / max(t2 / 2, 1e-6))))
    reps, done, t0 = 0, 0, time.perf_counter()
    while True:
        run(nb)
        reps += 1; done += nb
        dt = time.perf_counter() - t0
        if dt > budget_s or reps >= 50:
            break
    return {"value": done * args.points / dt, "unit": "points/s", "cores": ncores, "kind": "port",
            "sample": "%d x %d clouds of the workload (N=%d, L=%d): torch-CPU fp32 flow oracle on %d threads + "
                      "single-thread C Chamfer oracle, %.1f s" % (reps, nb, args.points, args.layers, ncores, dt)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    n_gpus = world if world > 1 else 1

    dec, state, n_flows, z, g, tgt, tgt_pm = build_workload(args, device)
    L = args.layers
    step = make_step(dec, z, g, tgt_pm, L)

    # first calls: pack weights, set LDS attribute -- outside any capture and outside the timed region
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()

    # Every step is one full pass over one batch.  With --streams S > 1 consecutive steps go round-robin over S streams, each
    # stream with its own captured graph and therefore its own intermediate / output buffers, so up to S steps are in flight
    # and the Chamfer kernels of one share the chip with the flow kernel of the next; all K steps have finished at the
    # closing synchronize.
    S = max(1, args.streams)
    S_all = max(S, args.pipelined if S == 1 else 0)
    streams = [torch.cuda.Stream() for _ in range(S_all)]
    runners = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            out = step()
        torch.cuda.synchronize()
        if args.no_graph:
            runners.append(step)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                out = step()
            runners.append(graph.replay)

    def run_steps(n, S=S):
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                runners[i % S]()

    run_steps(args.settle)
    run_steps(args.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pipelined = None
    if S_all > S:                        # the same steps with S_all of them in flight, reported beside the headline
        run_steps(args.warmup, S_all)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        run_steps(args.steps, S_all)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - tp) / args.steps

    # sanity on the outputs of the timed path (cheap, outside the timed region)
    p_out, d1, i1, d2, i2, cd = out
    assert torch.isfinite(p_out).all() and torch.isfinite(cd).all() and (d1 >= 0).all()

    if rank == 0:
        pts_per_step = args.batch * args.points * n_gpus
        ms_per_step = elapsed / args.steps * 1e3
        kt_alone = kernel_timings(dec, z, g, tgt_pm, L, args.precision)
        kt = kernel_timings_in_flight(dec, z, g, tgt_pm, L, args.precision, S) if S > 1 else kt_alone
        from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
        dom = max(("flow_kernel", "nn_kernel"), key=lambda k: kt[k])
        B, N = args.batch, args.points
        flow_flops = FLOP_PER_POINT_LAYER * L * B * N
        flow_ach = flow_flops / (kt["flow_kernel"] * 1e-6) / 1e12
        nn_bytes = B * (N + N) * 20                         # 12 B in + 8 B out per point
        nn_ach = nn_bytes / (kt["nn_kernel"] * 1e-6) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                tr = json.load(open(tj))
                key = "%s/B%d_N%d_L%d_%s" % (dom, B, N, L, args.precision)
                traffic = (tr.get(key) or {}).get("hbm_bytes")
            except Exception:
                traffic = None
        if dom == "flow_kernel":
            roof = {"kernel": "flow_kernel<%s>" % args.precision, "bound": "mfma", "achieved": flow_ach,
                    "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flow_ach / MFMA_BF16_PEAK_TF,
                    "traffic": traffic,
                    "note": "algorithmic FLOPs (17152/pt/layer); the split precision issues %dx the products of "
                            "the 64x64 contraction on the matrix cores" % MFMA_PRODUCTS[args.precision]}
        else:
            roof = {"kernel": "nn_kernel", "bound": "hbm", "achieved": nn_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nn_ach / HBM_PEAK_GBS, "traffic": traffic,
                    "note": "exact Chamfer is compute bound (AI ~800 FLOP/B); HBM fraction is tiny by construction"}
        roof["kernels_us"] = kt
        roof["kernels_us_alone"] = kt_alone
        if S > 1:
            roof["note"] += "; durations are those of the timed regime (%d steps in flight share the chip: each kernel takes " \
                            "longer than alone, the steps take less), kernels_us_alone = each kernel by itself" % S
        roof["flow_algorithmic_tflops"] = flow_ach
        roof["chamfer_algorithmic_gbs"] = nn_ach
        roof["chamfer_pair_evals_per_s"] = 2.0 * B * N * N / (kt["nn_kernel"] * 1e-6)
        line = {
            "metric": "points/sec through %d-layer flow + Chamfer, B=%d N=%d" % (L, B, N),
            "value": pts_per_step / (elapsed / args.steps), "unit": "points/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision + " MFMA operands, fp32 accumulate/points",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d coupling layers (first %d of LocalCondRNVPDecoder(n_flows=%d)), "
                                   "direct/eval-BN, + nn_distance both directions + CD reduction" % (L, L, n_flows),
                       "clouds_per_gpu": B, "points_per_cloud": N, "hidden": 64, "latent": args.latent,
                       "global_clouds": B * n_gpus, "per_layer_lists": bool(args.lists),
                       "launch": ("eager" if args.no_graph else "hipGraph replay") +
                       (", consecutive steps round-robin over %d streams with their own buffers" % S if S > 1 else ""),
                       "steps_in_flight": S, "settle_steps": args.settle, "parallelism": "clouds sharded, no collective",
                       "chamfer_impl": BK.NN_IMPL + (" (matrix-core filtered exact search at this size)" if BK.NN_IMPL == "auto" and
                                                     2.0 * B * N * N >= 1e8 and B * 2 * ((N + 255) // 256) >= 64 else "")},
            "roofline": roof,
        }
        if pipelined is not None:
            line["pipelined"] = {
                "steps_in_flight": S_all, "value": args.batch * args.points / pipelined,
                "unit": "points/s per GPU", "ms_per_step": pipelined * 1e3,
                "kernels_us": kernel_timings_in_flight(dec, z, g, tgt_pm, L, args.precision, S_all),
                "note": "consecutive steps round-robin over %d streams with their own buffers (bench.py --streams %d makes this "
                        "the timed regime): the Chamfer kernels of one step share the chip with the flow kernel of the next, "
                        "each kernel takes longer than alone, the steps take less" % (S_all, S_all)}
        if not args.no_cpu_baseline and n_gpus == 1:      # the CPU oracle is timed at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args, state, n_flows, tgt)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
