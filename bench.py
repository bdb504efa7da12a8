#!/usr/bin/env python3
"""Headline benchmark: points/s through the L-layer per-point flow + Chamfer.

  python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--layers 14|15|63] [--leg eval|train]
                  [--model decoder|encoder|autoencoder]

One "step" (leg eval, the BASELINE metric) = one pass of the hot path over one synthetic batch resident in HBM:
  FiLM conditioner (all layers) -> fused L-layer coupling stack, mode 'direct', eval-BN
  (lib/networks/decoders.py:54-72 as evaluate() runs it, evaluating.py:70)
  -> nn_distance(out^T, target) both directions (evaluating.py:110-111) -> per-cloud CD reduction (evaluating.py:112).
Workloads (BASELINE.json `configs`):
  cfg2 (default)  configs[1]: B=32 clouds x N=2048 per GPU, G=128, L=14 (first 14 layers of n_flows=5); weak scaling
  cfg3            configs[2]: all-classes model, 64 clouds in total sharded over the ranks, G=512; strong scaling
  cfg4            configs[3]: single-view reconstruction shapes (G=512, B=32 per GPU) + the f_score pass; weak scaling
  cfg5            configs[4]: 16 clouds of N=M=8192 in total, nn_distance + match_cost (approx-EMD); strong scaling
Leg train (also reported as `extra.train_step` of the default run): inverse stack in training mode (batch-statistics
BatchNorm) + PointFlowNLL + backward on a flattened decoder -> ONE all-reduce of the model's flat gradient (RCCL; counted) ->
Adam (lib/networks/training.py:37-56 with the collective between :55 and :56); `--model autoencoder` = the whole
Local_Cond_RNVP_MC_Global_RNVP_VAE of the workload's YAML (cfg3: 12 972 413 parameters, one 51.9 MB message).  The leg first
runs 6 steps with graph replay and 6 with eager launches and refuses to report a value if their losses differ.
The default run also carries `extra.configs`: short legs of L = 63 / 15, cfg3, cfg4 and cfg5 measured in the same process.

N GPUs: one process per GPU.  `--gpus N` without a torchrun environment starts the N ranks itself (children are
created before this process touches the GPU); under `python -m torch.distributed.run` the ranks read
RANK/LOCAL_RANK/WORLD_SIZE.  value = all ranks' points / max-over-ranks time.  Rank 0 prints ONE JSON line carrying
`roofline`, `cpu_baseline` and `parity`.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak
XGMI_LINK_GBS = 153.0          # per link and direction; 7 links per GPU
FLOP_PER_POINT_LAYER = 17152   # 2 branches x (2F|K| + 2F^2 + 2|W|F)
MFMA_PRODUCTS = {"bf16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}
CONFIGS = {
    #        clouds, per-GPU?, points, latent, scaling
    "cfg2": dict(clouds=32, per_gpu=True, points=2048, latent=128, scaling="weak",
                 name="configs[1]: airplane autoencoder shapes, B=32 N=2048 per GPU"),
    "cfg3": dict(clouds=64, per_gpu=False, points=2048, latent=512, scaling="strong",
                 name="configs[2]: all-classes autoencoder (all_scaled), 64 clouds sharded over the ranks, G=512"),
    "cfg4": dict(clouds=32, per_gpu=True, points=2048, latent=512, scaling="weak",
                 name="configs[3]: single-view reconstruction shapes (train_all_svr: G=512), B=32 N=2048 per GPU; the step adds the "
                      "f_score of evaluating.py:201-203 (a second nn_distance + thresholds); the ResNet image encoder stays on "
                      "PyTorch-ROCm and is not part of the hot path"),
    "cfg5": dict(clouds=16, per_gpu=False, points=8192, latent=128, scaling="strong",
                 name="configs[4]: dense clouds N=M=8192, 16 clouds sharded over the ranks, Chamfer + approx-EMD"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip needs a few hundred steps (tens of ms) under load to settle its clocks -- with 20 warm-up steps the
    # same step measures 98-99 us, after 300+ it measures 93 us and stays there
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--leg", default="eval", choices=["eval", "train"])
    ap.add_argument("--batch", type=int, default=None, help="clouds per GPU (overrides the config)")
    ap.add_argument("--strong", action="store_true",
                    help="STRONG scaling of a per-GPU config (cfg2 / cfg4): the config's 32 clouds are the GLOBAL batch, sharded "
                         "over the ranks (4 per rank at 8 GPUs) -- what an 8-GPU job does to BASELINE's B=32; the default line is "
                         "weak scaling (32 clouds per rank)")
    ap.add_argument("--no-proxy", action="store_true", help="skip `extra.per_rank_proxy` of the default run")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None, help="coupling layers: 14 (BASELINE metric), 15, 63; train leg default 63")
    ap.add_argument("--latent", type=int, default=None)
    ap.add_argument("--precision", default=os.environ.get("DPF_PRECISION", "f16x3"), choices=sorted(MFMA_PRODUCTS))
    ap.add_argument("--lists", action="store_true", help="also materialise the 3 x L per-layer lists")
    ap.add_argument("--settle", type=int, default=None,
                    help="untimed steps run BEFORE the --warmup steps so that the chip's clocks have settled under this load "
                         "whatever --warmup is; default 400 for the eval leg of cfg2/cfg3, 0 otherwise")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph-steps", type=int, default=10,
                    help="steps captured per hipGraph (a replay has a fixed cost of ~10 us whatever it holds); n steps run as "
                         "n // G replays of the G-step graph + n %% G replays of a one-step graph, so exactly n steps execute")
    ap.add_argument("--streams", type=int, default=1,
                    help="consecutive steps go round-robin over this many streams (each with its own buffers); 1 (default) = "
                         "strictly one after another, the regime the roofline numbers and the committed profiles describe")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="with --streams 1: also report (outside the timed region, as `pipelined`) the throughput with this many "
                         "steps in flight; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs of the default run (bf16x6, eager, train step)")
    ap.add_argument("--no-configs", action="store_true", help="skip `extra.configs` (the short legs of the other workloads) of the default run")
    ap.add_argument("--train-steps", type=int, default=20, help="timed steps of the `extra.train_step` leg")
    ap.add_argument("--model", default=None, choices=["decoder", "encoder", "autoencoder"],
                    help="train leg: decoder = the flow decoder on given codes (default), encoder = PointNet encoder + code head + "
                         "decoder, autoencoder = the whole Local_Cond_RNVP_MC_Global_RNVP_VAE of the workload's YAML (cfg3: "
                         "12 972 413 parameters, the 52 MB message of SURVEY 8e)")
    ap.add_argument("--encoder", default="none", choices=["none", "hip", "tensor"],
                    help="train leg: put the training-mode PointNet encoder (+ a linear code head) in front of the decoder, as "
                         "models.py:130-140 does; hip = csrc/encoder_train.hip, tensor = the tensor-op path")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = {"cfg5": 10}.get(args.config, 1000) if args.leg == "eval" else 40
    if args.warmup is None:
        args.warmup = {"cfg5": 3}.get(args.config, 300) if args.leg == "eval" else 12   # train: the allocator settles and the stack's calls become graph replays within ~10 steps
    if args.settle is None:
        args.settle = 400 if (args.leg == "eval" and args.config != "cfg5") else 0
    if args.points is None:
        args.points = cfg["points"]
    if args.latent is None:
        args.latent = cfg["latent"]
    if args.layers is None:
        args.layers = 14 if args.leg == "eval" else 63
    return args


# ----------------------------------------------------------------------------------------------------------------
# launching the ranks
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Watchdog:
    """Turns a rank that stops making progress into a diagnosis (profiles/README.md r05: one of seven two-rank runs
    hung in its first all-reduce -- one rank's step never reached the collective -- and nothing in the output could say where).
    Every rank runs one: the legs call beat(stage) at every step and either side of every collective; a daemon thread that sees
    no beat for DPF_BENCH_WATCHDOG_S seconds (default 240; 0 = off) prints ONE JSON line on stderr -- rank, the last stage, seconds
    without progress, the training engine's counters (graph replays / eager calls / recordings / evictions / uncapturable keys,
    workgroups that gave up waiting for their launch's role workgroups) -- and ends THIS process with status 86.  It never
    re-executes anything: torchrun (or launch_ranks below, the parent, which has not touched the GPU) then tears the other
    ranks down and the job exits non-zero with that line in its log."""
    _one = None

    def __init__(self, rank, seconds):
        import threading
        self.rank, self.seconds, self.stage, self.t, self.beats = rank, seconds, "start", time.monotonic(), 0
        self.lib_handle = None
        self.measured = None      # rank 0's headline line once its leg is done: a stall in an `extra` must not lose it
        if seconds > 0:
            threading.Thread(target=self._watch, name="bench-watchdog", daemon=True).start()

    @classmethod
    def start(cls, rank):
        cls._one = cls(rank, float(os.environ.get("DPF_BENCH_WATCHDOG_S", "240")))
        return cls._one

    @classmethod
    def beat(cls, stage):
        w = cls._one
        if w is not None:
            w.stage, w.t, w.beats = stage, time.monotonic(), w.beats + 1

    @classmethod
    def keep(cls, line):
        """rank 0: the line measured so far (None once main() has printed it itself)."""
        if cls._one is not None:
            cls._one.measured = line

    def _watch(self):
        while True:
            time.sleep(min(1.0, self.seconds / 4))
            idle = time.monotonic() - self.t
            if idle > self.seconds:
                dump = {"watchdog": "no progress", "rank": self.rank, "last_stage": self.stage, "beats": self.beats,
                        "seconds_without_progress": round(idle, 1), "pid": os.getpid()}
                try:                                          # host-side counters only: no GPU call from this thread
                    h = self.lib_handle
                    if h is not None:
                        st = (ctypes.c_long * 5)()
                        h.dpf_train_graph_stats(st)
                        dump["train_graph_stats"] = dict(zip(("replays", "eager", "recordings", "evictions", "uncapturable"), list(st)))
                        h.dpf_train_colsum_fallbacks.restype = ctypes.c_long
                        dump["colsum_fallbacks"] = int(h.dpf_train_colsum_fallbacks())
                except Exception as e:       # noqa: BLE001
                    dump["counters_error"] = repr(e)
                sys.stderr.write(json.dumps(dump) + "\n")
                sys.stderr.flush()
                line, self.measured = self.measured, None
                if line is not None:       # the headline was measured before the stall: print it, marked, and still exit 86
                    line = dict(line)
                    line["extra"] = dict(line.get("extra") or {}, watchdog=dump,
                                         note="the job stalled in an extra leg AFTER this line's timed region; the later extras are lost")
                    sys.stdout.write(json.dumps(line) + "\n")
                    sys.stdout.flush()
                os._exit(86)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children of THIS process, which has not touched the
    GPU (no HIP call so far: torch.cuda.device_count() does not initialise it), wait, and exit with their code."""
    if os.environ.get("DPF_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    # the ranks police themselves (Watchdog: status 86 + one JSON line); the parent adds the outer bound: a job that is still
    # there after DPF_BENCH_LAUNCH_TIMEOUT_S (default 3600) has its whole process group killed -- by this process, which has never
    # initialised the GPU -- and the launcher exits non-zero
    limit = float(os.environ.get("DPF_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit if limit > 0 else None)
    except subprocess.TimeoutExpired:
        import signal
        sys.stderr.write(json.dumps({"launcher": "timeout", "seconds": limit, "action": "killing the ranks (exact PIDs: the tree below "
                                     "the launcher's child)"}) + "\n")
        victims = []
        try:                                   # torchrun gives its workers their own process groups: walk the tree
            import psutil
            victims = psutil.Process(proc.pid).children(recursive=True)
        except Exception:       # noqa: BLE001
            pass
        for v in victims:
            try:
                v.kill()
            except Exception:       # noqa: BLE001
                pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        return 124


def init_ranks(args):
    """-> (rank, local_rank, world, dist or None).  Fails loudly if the launcher's world size is not --gpus."""
    if "WORLD_SIZE" not in os.environ:
        return 0, 0, 1, None
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, local_rank, 1, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("DPF_BENCH_BACKEND", "nccl")     # "gloo": the launcher/selftest path of the CPU tests
    import datetime
    # a collective nobody else joins must end the job in minutes, not after the default half hour
    pg_timeout = datetime.timedelta(seconds=int(os.environ.get("DPF_BENCH_PG_TIMEOUT", "300")))
    if backend == "nccl":
        # RCCL's own account of what it built (transports, channels, rings / trees) goes to a per-rank file and its first lines into
        # `extra.rccl` of rank 0's line, so that the first real multi-GPU run explains itself.  INIT lines only by default (nothing is
        # printed per collective); DPF_BENCH_RCCL_TUNING=1 adds the per-call algorithm / protocol choice (a host-side print per step).
        if os.environ.get("DPF_BENCH_RCCL_LOG", "1") != "0":
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING" if os.environ.get("DPF_BENCH_RCCL_TUNING") == "1" else "INIT,GRAPH")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/dpf_bench_rccl_%h_%p.log")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    else:
        dist.init_process_group(backend, timeout=pg_timeout)
    return rank, local_rank, world, dist


def rccl_log_excerpt(max_lines=24):
    """The informative lines of this process's RCCL log (NCCL_DEBUG_FILE, see init_ranks) or None."""
    import glob
    import re
    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return None
    try:
        files = [f for f in glob.glob(pat.replace("%h", "*").replace("%p", "*")) if f.endswith("_%d.log" % os.getpid())]
        keep = re.compile(r"(Channel \d+/\d+|Ring \d+|Tree|via |P2P|XGMI|xGMI|NET/|Connected all|comm 0x|nRanks|Algo|Proto|algo|proto|threadThresholds|RCCL version|NCCL version)")
        out = []
        for f in files:
            for l in open(f, errors="replace"):
                if keep.search(l):
                    out.append(l.strip()[-220:])
                    if len(out) >= max_lines:
                        break
            try:
                os.remove(f)                   # (the excerpt is in the line; no per-rank logs left behind in /tmp)
            except OSError:
                pass
        return out or None
    except Exception as e:       # noqa: BLE001
        return ["(could not read the RCCL log: %r)" % (e,)]


def timed_region(run_steps, args, dist, device, launch_streams=None):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; max over ranks.

    -> (seconds by HIP events, seconds by the host's clock).  The first is the number of record (hipEvents on the
    stream the kernels are launched on -- `launch_streams`, default torch's current stream); the second is the r01-r05 bracket,
    `perf_counter` around the same K steps and the closing synchronize, kept beside it as `ms_per_step_wall`.  The two differ by
    the host's share of a short window (one synchronize return + the first launch's latency, ~20 us: 1 us per step at K = 20).
    The garbage collector is held off for the window: a collection between two launches is a GPU idle gap that is not the
    path's (tools/headline_window.py measures what an idle gap in front of the window costs)."""
    import gc
    streams = list(launch_streams) if launch_streams else [torch.cuda.current_stream()]
    Watchdog.beat("timed region: warm-up")
    run_steps(args.warmup)
    Watchdog.beat("timed region: barrier in front of the timed steps")
    if dist is not None:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gc_was = gc.isenabled()
    gc.disable()
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(streams[0])               # every launch stream is idle here: one start stamp serves them all
        run_steps(args.steps)
        for st in streams[1:]:
            streams[0].wait_stream(st)
        ev1.record(streams[0])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        Watchdog.beat("timed region: timed steps done")
    finally:
        if gc_was:
            gc.enable()
    elapsed = ev0.elapsed_time(ev1) * 1e-3
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed, wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, wall = float(t[0].item()), float(t[1].item())
    return elapsed, wall


def device_clock_mhz():
    """The GPU's current shader clock (MHz) or None.  Context for a line, not part of any rate: r04 saw boxes of the pool run
    the same kernels 1.1-1.5x apart (VALU-bound kernels most), and the clock read right behind the timed loop tells which."""
    try:
        return int(torch.cuda.clock_rate())
    except Exception:
        return None


def clouds_of_rank(args, rank, world):
    cfg = CONFIGS[args.config]
    if args.batch is not None:
        return args.batch, args.batch * world
    if cfg["per_gpu"] and not getattr(args, "strong", False):
        return cfg["clouds"], cfg["clouds"] * world
    from dpf_nets_amd.distributed import shard_bounds
    lo, hi = shard_bounds(cfg["clouds"], rank, world)
    if hi - lo < 1:
        raise SystemExit("bench.py: %s has %d clouds, too few for %d ranks" % (args.config, cfg["clouds"], world))
    return hi - lo, cfg["clouds"]


# ----------------------------------------------------------------------------------------------------------------
# leg eval: flow (direct, eval-BN) + Chamfer
# ----------------------------------------------------------------------------------------------------------------
def build_workload(args, device, batch):
    from dpf_nets_amd import synthetic as FO
    from dpf_nets_amd.networks import LocalCondRNVPDecoder
    n_flows = (args.layers + 2) // 3
    state = FO.make_decoder_state(0, n_flows, 64, args.latent)
    dec = LocalCondRNVPDecoder(n_flows, 64, args.latent)
    dec.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}, strict=True)
    dec = dec.to(device).eval()
    dec.precision = args.precision
    dec.materialize_lists = bool(args.lists)
    tgt, z, g = FO.synthetic_inputs(0, batch, args.points, args.latent)
    z = torch.from_numpy(z).to(device)
    g = torch.from_numpy(g).to(device)
    tgt_pm = torch.from_numpy(np.ascontiguousarray(tgt.transpose(0, 2, 1))).to(device)   # (B,N,3), resident
    return dec, state, n_flows, z, g, tgt, tgt_pm


def make_step(dec, z, g, tgt_pm, L, precision=None, with_fscore=False):
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    from dpf_nets_amd.networks.utils import chamfer_per_cloud, f_score, ChamferEvaluator
    stack = dec.stack()
    precision = precision or dec.precision
    # the package's own evaluation-loop helper (networks.utils.ChamferEvaluator): it owns the scratch of the fused Chamfer + CD
    # call, one workspace per (shape, stream), made outside any capture -- the call is then ONE launch.  The benchmark measures
    # the path the package's evaluation helpers take (metrics.evaluation_metrics.EMD_CD, chamfer_cd_per_cloud), not a private one
    evaluator = ChamferEvaluator()

    def step():
        p_out, sum_lv, ps, mus, lvs = stack.run(z, g, "direct", precision, want_lists=dec.materialize_lists,
                                                n_layers=L, want_pointmajor=True)
        if BK.NN_IMPL == "auto":          # distances, indices and the per-cloud CD from the search kernel's workgroups
            d1, i1, d2, i2, cd = evaluator(stack.last_pointmajor, tgt_pm)
        else:
            d1, i1, d2, i2 = BK.NNDistance(stack.last_pointmajor, tgt_pm)
            cd = chamfer_per_cloud(d1, d2)
        if with_fscore:                   # evaluating.py:201-203 ('predicting' mode): a second nn_distance + thresholds
            cd = cd + 0.0 * f_score(stack.last_pointmajor, tgt_pm)
        return p_out, sum_lv, d1, i1, d2, i2, cd
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
    for _ in range(25):                  # settle the clocks under this kernel's load, as in the timed loop
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
    canon, meta, packed, G, precision = stack._ensure(precision, z.device, L)
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
    """The same durations in the regime of a pipelined loop: steps round-robin over `n_streams` streams, HIP events on the
    launch stream around each kernel (kernels of neighbouring steps share the chip, so each takes longer than alone while
    the steps take less)."""
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


def read_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass (collected by tools/pmc_run.sh
    with the same command, corrected as MI355X_MICROARCH.md prescribes) -- counters cannot be read inside this process."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tj):
        return None, None
    try:
        tr = json.load(open(tj))
        ent = tr.get(key) or {}
        return ent.get("hbm_bytes"), ent.get("source", "profiles/traffic.json")
    except Exception:
        return None, None


# ---- the CPU oracle: the checker (parity) and the reported baseline.  Nothing else in this file touches oracle/. -------
def cpu_baseline_and_parity(args, state, n_flows, batch, tgt, gpu_out, time_it, budget_s=15.0):
    """Runs the CPU oracle (kind 'port': torch-CPU fp32 restatement of flows.py:95-117 + the C restatement of
    nndistance.cu) on whole clouds of the workload.
    parity: the GPU outputs of the first two clouds against it -- flow: norm-wise and elementwise error; Chamfer: the GPU
    Chamfer of the GPU flow output against the C oracle fed the same (GPU) flow output, bit for bit.
    cpu_baseline (time_it): as many clouds as fit ~budget_s of CPU work, timed."""
    from oracle import flow_oracle as FO
    from oracle import structural as S
    ncores = min(os.cpu_count() or 1, 32)       # torch-CPU on hundreds of threads thrashes on these small ops
    torch.set_num_threads(ncores)
    st = FO.to_torch(state)
    _, z, g = FO.synthetic_inputs(0, batch, args.points, args.latent)
    tgt_pm = np.ascontiguousarray(tgt.transpose(0, 2, 1))
    S.lib()
    S.set_threads(ncores)                       # both legs on the same threads (the queries of the C restatement are independent)

    def flow(nb):
        tz, tg = torch.from_numpy(z[:nb]), torch.from_numpy(g[:nb])
        with torch.no_grad():
            ps, _, lvs = FO.decoder(st, n_flows, tz, tg, "direct", n_layers=args.layers)
        return ps[-1].numpy(), sum(lvs).numpy()

    def run(nb):
        out, _ = flow(nb)
        d1, _, d2, _ = S.nndistance(np.ascontiguousarray(out.transpose(0, 2, 1)), tgt_pm[:nb])
        return d1.mean(1) + d2.mean(1)

    nchk = min(2, batch)
    ref_p, ref_lv = flow(nchk)
    p_out, sum_lv, d1, i1, d2, i2, cd = [t[:nchk].cpu().numpy() for t in gpu_out]

    def errs(got, ref):
        scale = float(np.abs(ref).max())
        e = np.abs(got - ref)
        return float(e.max() / scale), float(e.max()), int((e > 1e-4 * np.abs(ref) + 1e-5 * scale).sum())
    rel_p, abs_p, bad_p = errs(p_out, ref_p)
    rel_l, abs_l, bad_l = errs(sum_lv, ref_lv)
    r1, j1, r2, j2 = S.nndistance(np.ascontiguousarray(p_out.transpose(0, 2, 1)), tgt_pm[:nchk])
    parity = {"precision": args.precision, "sample": "%d clouds x %d points, L=%d" % (nchk, args.points, args.layers),
              "max_rel_vs_oracle": max(rel_p, rel_l), "max_abs_elementwise": max(abs_p, abs_l),
              "elementwise_violations_rtol1e-4_atol1e-5scale": bad_p + bad_l,
              "points_max_rel": rel_p, "sum_logvar_max_rel": rel_l,
              "chamfer_bit_exact": bool(np.array_equal(d1.view(np.uint32), r1.view(np.uint32)) and np.array_equal(i1, j1) and
                                        np.array_equal(d2.view(np.uint32), r2.view(np.uint32)) and np.array_equal(i2, j2))}
    base = None
    if time_it:
        run(1)                                       # warm-up
        t0 = time.perf_counter(); run(2); t2 = time.perf_counter() - t0          # calibration on 2 clouds
        nb = int(max(2, min(batch, budget_s / max(t2 / 2, 1e-6))))
        reps, done, t0 = 0, 0, time.perf_counter()
        while True:
            run(nb)
            reps += 1; done += nb
            dt = time.perf_counter() - t0
            if dt > budget_s or reps >= 50:
                break
        base = {"value": done * args.points / dt, "unit": "points/s", "cores": ncores, "kind": "port",
                "sample": "%d x %d clouds of the workload (N=%d, L=%d): torch-CPU fp32 flow oracle + C Chamfer oracle "
                          "(OpenMP over queries), both on %d threads, %.1f s" % (reps, nb, args.points, args.layers, ncores, dt)}
    return base, parity


def leg_eval(args, rank, world, dist, device):
    batch, global_clouds = clouds_of_rank(args, rank, world)
    dec, state, n_flows, z, g, tgt, tgt_pm = build_workload(args, device, batch)
    L = args.layers
    step = make_step(dec, z, g, tgt_pm, L, with_fscore=args.config == "cfg4")
    # first calls: pack weights, set LDS attribute -- outside any capture and outside the timed region
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()

    # Every step is one full pass over one batch.  With --streams S > 1 consecutive steps go round-robin over S streams, each
    # stream with its own captured graph and therefore its own intermediate / output buffers.
    S = max(1, args.streams)
    S_all = max(S, args.pipelined if S == 1 else 0)
    streams = [torch.cuda.Stream() for _ in range(S_all)]
    G = max(1, args.graph_steps) if (S == 1 and not args.no_graph) else 1     # multi-step graphs only in the one-stream regime
    runners, multi = [], None
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
            if G > 1 and multi is None:
                multi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(multi, stream=st):
                    for _ in range(G):
                        out = step()

    def run_steps(n, S=S):
        if multi is not None and S == 1:
            with torch.cuda.stream(streams[0]):
                for _ in range(n // G):
                    multi.replay()
                for _ in range(n % G):
                    runners[0]()
            return
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                runners[i % S]()

    run_steps(args.settle)
    elapsed, elapsed_wall = timed_region(run_steps, args, dist, device, launch_streams=streams[:S])
    sclk = device_clock_mhz()            # right behind the timed region: the shader clock the steps ran at (boxes differ)

    pipelined = None
    if S_all > S:                        # the same steps with S_all of them in flight, reported beside the headline
        run_steps(args.warmup, S_all)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        run_steps(args.steps, S_all)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - tp) / args.steps

    p_out, sum_lv, d1, i1, d2, i2, cd = out     # sanity on the outputs of the timed path (outside the timed region)
    assert torch.isfinite(p_out).all() and torch.isfinite(cd).all() and (d1 >= 0).all()

    extra = {}
    if not args.no_extra and not args.no_graph and S == 1:
        extra = eval_extras(args, dec, z, g, tgt_pm, L, step, streams[0])
    if rank != 0:
        return None, extra
    B, N = batch, args.points
    pts_per_step = args.points * global_clouds
    kt_alone = kernel_timings(dec, z, g, tgt_pm, L, args.precision)
    kt = kernel_timings_in_flight(dec, z, g, tgt_pm, L, args.precision, S) if S > 1 else kt_alone
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    from dpf_nets_amd._lib import lib as _dpf_lib
    n16 = _dpf_lib().dpf_flow_tile16_launches()
    step()
    torch.cuda.synchronize()
    tile16 = _dpf_lib().dpf_flow_tile16_launches() > n16      # which tiling of the fused stack serves this batch size
    flow_name = ("flow16_kernel<%s> (16-point tiles, csrc/flow16.hip)" if tile16 else "flow_kernel<%s>") % args.precision
    dom = max(("flow_kernel", "nn_kernel"), key=lambda k: kt[k])
    flow_flops = FLOP_PER_POINT_LAYER * L * B * N
    flow_ach = flow_flops / (kt["flow_kernel"] * 1e-6) / 1e12
    nn_bytes = B * (N + N) * 20                         # 12 B in + 8 B out per point
    nn_ach = nn_bytes / (kt["nn_kernel"] * 1e-6) / 1e9
    traffic, tsrc = read_traffic("%s/B%d_N%d_L%d_%s" % (dom, B, N, L, args.precision))
    if dom == "flow_kernel":
        roof = {"kernel": flow_name, "bound": "mfma", "achieved": flow_ach,
                "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flow_ach / MFMA_BF16_PEAK_TF,
                "traffic": traffic,
                "note": "algorithmic FLOPs (17152/pt/layer) per launch / HIP-event launch duration; the split precision "
                        "issues %dx the products of the 64x64 contraction on the matrix cores" % MFMA_PRODUCTS[args.precision]}
    else:
        roof = {"kernel": "nn_kernel", "bound": "hbm", "achieved": nn_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": nn_ach / HBM_PEAK_GBS, "traffic": traffic,
                "note": "exact Chamfer is compute bound (AI ~800 FLOP/B); HBM fraction is tiny by construction"}
    roof["traffic_source"] = tsrc and (tsrc + " (rocprofv3 --pmc pass of this command, not measured in this run)")
    roof["kernels_us"] = kt
    roof["kernels_us_alone"] = kt_alone
    # flat copies (the driver's parser keeps scalars of `roofline` / `config`, not nested objects): with `config.sclk_mhz` and
    # `config.ms_per_step_wall` they tell a slow box (every kernel and the step up together) from a regression (one kernel up)
    # from a gap in the window (step up, kernels_sum_us not)
    roof["flow_us"], roof["nn_us"], roof["film_us"] = kt["flow_kernel"], kt["nn_kernel"], kt["film_kernel"]
    roof["kernels_sum_us"] = kt["flow_kernel"] + kt["nn_kernel"] + kt["film_kernel"]
    if S > 1:
        roof["note"] += "; durations are those of the timed regime (%d steps in flight share the chip)" % S
    roof["flow_algorithmic_tflops"] = flow_ach
    roof["chamfer_algorithmic_gbs"] = nn_ach
    roof["chamfer_pair_evals_per_s"] = 2.0 * B * N * N / (kt["nn_kernel"] * 1e-6)
    # the meaningful Chamfer fraction is pair evaluations against the fp32 VALU peak, 8 FLOP per pair for the
    # brute-force formula (3 sub, 3 mul, 2 add) -- an EQUIVALENT rate: the default kernel gets most pairs out of the way with one
    # bf16 MFMA per 32x32 of them and evaluates the exact formula only for the candidates that can still win
    roof["chamfer_equiv_fp32_tflops"] = 8.0 * roof["chamfer_pair_evals_per_s"] / 1e12
    roof["chamfer_equiv_frac_of_fp32_valu_peak"] = roof["chamfer_equiv_fp32_tflops"] / 157.3
    cfg = CONFIGS[args.config]
    line = {
        "metric": "points/sec through %d-layer flow + Chamfer, B=%d N=%d" % (L, B, N),
        "value": pts_per_step / (elapsed / args.steps), "unit": "points/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": ("strong" if getattr(args, "strong", False) else cfg["scaling"]) if args.batch is None else "weak", "vs_baseline": None,
        "dtype": args.precision + " MFMA operands, fp32 accumulate/points",
        "data": "synthetic",
        "config": {"workload": "%s: %d coupling layers (first %d of LocalCondRNVPDecoder(n_flows=%d)), direct/eval-BN, + "
                               "nn_distance both directions + CD reduction" % (cfg["name"], L, L, n_flows),
                   "clouds_per_gpu": B, "points_per_cloud": N, "hidden": 64, "latent": args.latent,
                   "global_clouds": pts_per_step // N, "per_layer_lists": bool(args.lists), "flow_kernel": flow_name,
                   "launch": ("eager" if args.no_graph else "hipGraph replay, %d step(s) per graph" % G) +
                   (", consecutive steps round-robin over %d streams with their own buffers" % S if S > 1 else ""),
                   "steps_in_flight": S, "settle_steps": args.settle, "parallelism": "clouds sharded, no collective",
                   "timing": "HIP events on the launch stream around the K steps (ms_per_step); host clock around the same "
                             "steps + synchronize beside it (ms_per_step_wall)",
                   "ms_per_step_wall": elapsed_wall / args.steps * 1e3, "sclk_mhz": sclk,
                   "chamfer_impl": BK.NN_IMPL + (" (matrix-core filtered exact search at this size)" if BK.NN_IMPL == "auto" and
                                                 2.0 * B * N * N >= 1e8 and B * 2 * ((N + 255) // 256) >= 64 else "")},
        "roofline": roof,
        "device": {"name": torch.cuda.get_device_name(device), "sclk_mhz_behind_timed_region": sclk},
    }
    if pipelined is not None:
        line["pipelined"] = {
            "steps_in_flight": S_all, "value": batch * args.points / pipelined,
            "unit": "points/s per GPU", "ms_per_step": pipelined * 1e3,
            "note": "consecutive steps round-robin over %d streams with their own buffers (bench.py --streams %d makes this "
                    "the timed regime): the Chamfer kernels of one step share the chip with the flow kernel of the next" %
                    (S_all, S_all)}
    base, parity = cpu_baseline_and_parity(args, state, n_flows, batch, tgt, out,
                                           time_it=not args.no_cpu_baseline and world == 1)   # CPU timing at N = 1 only
    line["cpu_baseline"] = base
    line["parity"] = parity
    return line, extra


def eval_extras(args, dec, z, g, tgt_pm, L, step, stream):
    """Outside the timed region, every rank: (a) the same step launched eagerly (what a caller that cannot capture pays),
    (b) the other split precisions (bf16x3, bf16x6) beside the benched one."""
    out = {}
    try:
        for _ in range(300):                  # the chip's clocks settle over tens of ms: as many untimed steps as the headline
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            step()
        torch.cuda.synchronize()
        out["eager_ms_per_step"] = (time.perf_counter() - t0) / 1000 * 1e3
    except Exception as e:       # noqa: BLE001 -- an extra must never cost the headline line
        out["eager_error"] = repr(e)
    ref_points = step()[0].clone()
    for other in [p for p in ("bf16x3", "bf16x6") if p != args.precision]:
        try:
            st2 = make_step(dec, z, g, tgt_pm, L, precision=other)
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                o2 = st2()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                o2 = st2()
            with torch.cuda.stream(stream):
                for _ in range(100):
                    graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(300):
                    graph.replay()
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 300
            out[other] = {"ms_per_step": dt * 1e3, "value": z.shape[0] * z.shape[2] / dt, "unit": "points/s per GPU",
                          "max_rel_points_vs_%s" % args.precision: float((o2[0] - ref_points).abs().max() / ref_points.abs().max())}
        except Exception as e:       # noqa: BLE001
            out[other + "_error"] = repr(e)
    return out


# ----------------------------------------------------------------------------------------------------------------
# leg cfg5: Chamfer + approx-EMD on dense clouds
# ----------------------------------------------------------------------------------------------------------------
def leg_cfg5(args, rank, world, dist, device):
    from dpf_nets_amd import synthetic as FO
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    from dpf_nets_amd.networks.utils import chamfer_per_cloud
    batch, global_clouds = clouds_of_rank(args, rank, world)
    N = args.points
    tgt, z, _ = FO.synthetic_inputs(0, batch, N, 16)
    a = torch.from_numpy(np.ascontiguousarray(tgt.transpose(0, 2, 1))).to(device)
    # second cloud: the target jittered and permuted, so that the matching is non-trivial but of the same extent
    rng = np.random.default_rng(1)
    perm = rng.permutation(N)
    b_np = (tgt.transpose(0, 2, 1)[:, perm] + 0.02 * rng.standard_normal((batch, N, 3))).astype(np.float32)
    b = torch.from_numpy(np.ascontiguousarray(b_np)).to(device)

    def step():
        d1, i1, d2, i2 = BK.NNDistance(a, b)
        cd = chamfer_per_cloud(d1, d2)
        match, temp, cost = BK.ApproxMatchCost(a, b)
        return d1, i1, d2, i2, cd, cost

    out = step()
    torch.cuda.synchronize()

    def run_steps(n):
        for _ in range(n):
            step()
    elapsed, elapsed_wall = timed_region(run_steps, args, dist, device)
    if rank != 0:
        return None, {}

    def ev_time(fn, reps=5):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); e.synchronize()
        return s.elapsed_time(e) / reps * 1e3
    t_nn = ev_time(lambda: BK.NNDistance(a, b))
    t_emd = ev_time(lambda: BK.ApproxMatchCost(a, b))
    # Roofline of the dominant kernels (approx-EMD).  SURVEY 8(d) prices the REFERENCE's algorithm at 80*n*m B per cloud (the
    # (B,m,n) `match` read-modified-written on each of 9 levels): this implementation keeps the level state in a workspace
    # and writes `match` once, so it does not move those bytes -- dividing them by the time measured nothing (r02: a fraction
    # of 1.56).  What bounds it is the transcendental issue rate: 36 v_exp_f32 per pair (27 level passes + the materialise /
    # cost pass).  A wave64 v_exp_f32 occupies its SIMD for 8.1-8.5 cycles on this part (profiles/r02_mfma_fill.txt, rows
    # `exp`): 8 lanes per SIMD-clock, half the plain-VALU rate.  The quarter-rate figure VERDICT r02 #8 priced it at is
    # reported beside it.  (r03's first lines carried 256*4*16*2.4e9 here -- the FULL VALU lane rate, a slip: frac 0.14.)
    # The HBM side is reported against what the algorithm must write: 4*n*m B per cloud (`match`, once) -- counter bytes
    # from the committed --pmc pass, when present.
    # r05: the level passes run on the matrix cores over the LIVE part of cloud 2 only (csrc/emd.hip: points whose remainR has
    # reached 0 are dropped from the lists, ~2.1 level-passes' worth of pairs instead of 9 on these clouds), so the op no longer
    # issues the reference's 36 exp per pair and the exp rate is not what bounds it: the largest kernel is the materialisation,
    # which writes `match` (4*n*m B per cloud) once -- the roofline is that write against the HBM peak, over the whole call.
    # The reference-equivalent exp rate (36 per pair / time) is kept as a derived figure for comparison with r02-r04.
    exp_total = 36.0 * batch * N * N
    exp_peak = 256 * 4 * 8 * 2.4e9                      # CUs x SIMDs x 8 lanes per clock x 2.4 GHz = 1.966e13 exp/s
    ach = exp_total / (t_emd * 1e-6)
    must_write = 4.0 * batch * N * N
    traffic, tsrc = read_traffic("approxmatch_cost/B%d_N%d" % (batch, N))
    roof = {"kernel": "approxmatch + matchcost (dpf_approxmatch_cost_ws: 27 level passes over the live lists + 9 compactions + "
                      "1 materialise/cost pass; dominant kernel emd_mfma_materialize_kernel)",
            "bound": "hbm", "achieved": must_write / (t_emd * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": must_write / (t_emd * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes": must_write,
            "counter_bytes_over_algorithmic": (traffic / must_write) if traffic else None,
            "note": "algorithmic bytes = 4*n*m per cloud: `match` written once (the reference's own RMW form would move 80*n*m, "
                    "SURVEY 8d); achieved = those bytes / the whole approxmatch+cost call.  The materialisation kernel alone "
                    "streams them out at ~5.1 TB/s (nontemporal stores; profiles/r06_emd_kernel_trace.txt); the level passes before it are bound by "
                    "v_exp_f32 issue on the live pairs",
            "reference_equivalent_exp": {"per_pair": 36, "Gexp_per_s": ach / 1e9, "issue_peak_Gexp_per_s": exp_peak / 1e9,
                                         "note": "36 exp per pair is what the reference's 27 dense passes + materialisation "
                                                 "evaluate; this implementation skips the pairs whose weight is exactly 0"},
            "kernels_us": {"nn_distance": t_nn, "approxmatch_cost": t_emd},
            "chamfer_pair_evals_per_s": 2.0 * batch * N * N / (t_nn * 1e-6),
            "emd_exp_per_s": ach}
    line = {"metric": "points/sec through Chamfer + approx-EMD, B=%d N=M=%d" % (batch, N),
            "value": global_clouds * N / (elapsed / args.steps), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.batch is None else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": CONFIGS["cfg5"]["name"] + ": nn_distance both directions + CD reduction + match_cost "
                                   "(approxmatch + matchcost)", "clouds_per_gpu": batch, "points_per_cloud": N,
                       "global_clouds": global_clouds, "launch": "eager", "parallelism": "clouds sharded, no collective",
                       "timing": "HIP events on the launch stream (ms_per_step); host clock beside it (ms_per_step_wall)",
                       "ms_per_step_wall": elapsed_wall / args.steps * 1e3},
            "roofline": roof}
    base = parity = None
    if world == 1:
        from oracle import structural as S     # the checker / CPU baseline leg
        an, bn = a[:1].cpu().numpy(), b[:1].cpu().numpy()
        t0 = time.perf_counter()
        r1, j1, r2, j2 = S.nndistance(an, bn)
        timed = not args.no_cpu_baseline
        if timed:
            m_ref, _ = S.approxmatch(an, bn)
            c_ref = S.matchcost(an, bn, m_ref)
        dt = time.perf_counter() - t0
        d1, i1, d2, i2, cd, cost = [t[:1].cpu().numpy() for t in out]
        parity = {"sample": "1 cloud of the workload (n=m=%d)" % N,
                  "chamfer_bit_exact": bool(np.array_equal(d1.view(np.uint32), r1.view(np.uint32)) and np.array_equal(i1, j1) and
                                            np.array_equal(d2.view(np.uint32), r2.view(np.uint32)) and np.array_equal(i2, j2))}
        if timed:
            parity["emd_cost_rel_err_vs_oracle"] = float(abs(cost[0] - c_ref[0]) / abs(c_ref[0]))
            parity["note"] = "approx-EMD oracle is parity-unpinned (the reference has no CPU path; oracle/structural_oracle.c header)"
            base = {"value": N / dt, "unit": "points/s", "cores": 1, "kind": "port",
                    "sample": "1 of the %d clouds (n=m=%d): single-thread C oracle nndistance + approxmatch + matchcost, %.1f s"
                              % (batch, N, dt)}
    line["cpu_baseline"] = base
    line["parity"] = parity
    return line, {}


# ----------------------------------------------------------------------------------------------------------------
# leg train: inverse stack (training-mode BN) + NLL + backward + ONE all-reduce + Adam
# ----------------------------------------------------------------------------------------------------------------
def _lib_handle():
    from dpf_nets_amd._lib import lib
    return lib()


def _train_precision():
    from dpf_nets_amd.networks import train_engine
    return train_engine.TRAIN_PRECISION


# the network kwargs of the reference's YAML files, per workload (configs/generation/airplane.yaml:27-52,
# configs/autoencoding/all_scaled.yaml:27-51; svr/all.yaml shares all_scaled's decoder / prior shapes)
MODEL_CONFIGS = {
    128: dict(deterministic=False, pc_enc_init_n_channels=3, pc_enc_init_n_features=64, pc_enc_n_features=[128, 256, 512],
              g_latent_space_size=128, g_prior_n_flows=7, g_prior_n_features=128, g_posterior_n_layers=1, p_latent_space_size=3,
              p_prior_n_layers=1, p_decoder_n_flows=21, p_decoder_n_features=64, p_decoder_base_type="free",
              p_decoder_base_var=-3.9551, pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0, util_mode="training"),
    512: dict(deterministic=False, pc_enc_init_n_channels=3, pc_enc_init_n_features=64, pc_enc_n_features=[128, 256, 512],
              g_latent_space_size=512, g_prior_n_flows=7, g_prior_n_features=128, g_posterior_n_layers=1, p_latent_space_size=3,
              p_prior_n_layers=1, p_decoder_n_flows=21, p_decoder_n_features=64, p_decoder_base_type="freevar",
              p_decoder_base_var=-3.5960, pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0, util_mode="training"),
}


def build_train_workload(args, rank, device, batch, layers, model_kind):
    """-> (step_fn factory inputs): modules, parameters, a closure computing the loss of one batch.
    model_kind: "decoder" (the flow decoder alone on given codes: the r01/r02 leg), "encoder" (PointNet encoder + a linear
    code head in front of it), "autoencoder" (the whole Local_Cond_RNVP_MC_Global_RNVP_VAE of the workload's YAML: encoder,
    posterior, latent prior flow, base-distribution net, decoder, the three loss terms of losses.py:37-51)."""
    from dpf_nets_amd import networks as nets, synthetic as SY
    n_flows, G, N = layers // 3, args.latent, args.points
    torch.manual_seed(0)                      # same initial weights on every rank (replicas)
    tgt, _, g = SY.synthetic_inputs(3 + rank, batch, N, G)          # every rank its own shard of clouds
    tp, tg = torch.from_numpy(tgt).to(device), torch.from_numpy(g).to(device)
    if model_kind == "autoencoder":
        cfg = dict(MODEL_CONFIGS[512 if G >= 512 else 128], g_latent_space_size=G, p_decoder_n_flows=n_flows)
        model = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE(**cfg).to(device).train()
        model.pc_encoder.hip_training = getattr(args, "encoder", "hip") != "tensor"
        stores = model.flatten_parameters()
        loss_fn = nets.Local_Cond_RNVP_MC_Global_RNVP_VAE_Loss(**cfg)

        def compute():
            return loss_fn(tp, tp, model(tp, tp))[0]                 # training.py:37,42
        what = ("whole autoencoder of the workload's YAML (PointNet encoder [%s] + posterior + %d-flow latent prior + base net + "
                "%d-layer decoder; pnll + gnll - gent)" % ("HIP" if model.pc_encoder.hip_training else "tensor ops",
                                                           cfg["g_prior_n_flows"], layers))
        return list(model.parameters()), compute, stores[0], what
    dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).to(device).train()
    store = dec.flatten_parameters()
    params = list(dec.parameters())
    enc = head = None
    if model_kind == "encoder":
        # models.py:130-140: cloud code = head(max over the points of pc_encoder(p_input)); the reference's head is a
        # FeatureEncoder (tensor ops there and here) -- its mean layer stands in for it
        enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512]).to(device).train()
        enc.hip_training = args.encoder == "hip"
        head = torch.nn.Linear(512, G).to(device)
        params += list(enc.parameters()) + list(head.parameters())
    pm, pl = torch.zeros(batch, 3, N, device=device), torch.full((batch, 3, N), -3.6, device=device)
    nll = nets.PointFlowNLL()

    def compute():
        code = tg if enc is None else head(torch.max(enc(tp), dim=2)[0])
        ps, mus, lvs = dec(tp, code, mode="inverse")
        return nll(ps + [tp], [pm] + mus, [pl] + lvs)                # models.py:169-171, losses.py:48
    what = ("" if enc is None else "PointNet encoder (training mode, %s) + code head + " % args.encoder) + \
        "inverse %d-layer stack (batch-stat BN) + PointFlowNLL" % layers
    return params, compute, store, what


def replay_equals_eager(args, rank, device, batch, layers, model_kind, n=6):
    """The first n optimizer steps of the leg, once with the stack's calls recorded / replayed as hipGraphs and once with
    eager launches (fresh model, same seed): loss bytes must agree.  Outside every timed region."""
    from dpf_nets_amd import networks as nets, distributed as D
    L_ = _lib_handle()
    seqs, replays = [], []
    for on in (1, 0):
        prev = L_.dpf_train_graph_set_enabled(on)
        try:
            r0 = int(L_.dpf_train_graph_replays())
            params, compute, store, _ = build_train_workload(args, rank, device, batch, layers, model_kind)
            arena = D.GradArena(params)
            opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
            losses = []
            for _ in range(n):
                arena.zero_grad()
                torch.manual_seed(1234 + len(losses))
                loss = compute()
                loss.backward()
                opt.step()
                losses.append(loss.detach())
            seqs.append(torch.stack(losses).cpu().numpy().tobytes())
            replays.append(int(L_.dpf_train_graph_replays()) - r0)
            del params, compute, store, arena, opt, loss
        finally:
            L_.dpf_train_graph_set_enabled(prev)
    return seqs[0] == seqs[1], replays[0]


def train_step_leg(args, rank, world, dist, device, batch, layers, steps, warmup, model_kind=None):
    from dpf_nets_amd import networks as nets
    from dpf_nets_amd import distributed as D
    if layers % 3:
        raise SystemExit("train leg: --layers must be a multiple of 3 (whole CondRealNVPFlow3DTriple's)")
    N, G = args.points, args.latent
    if model_kind is None:
        model_kind = getattr(args, "model", None) or ("encoder" if getattr(args, "encoder", "none") != "none" else "decoder")
    # Everything that can fail on ONE rank (allocation, the replay check) happens before the first collective, and the ranks
    # vote on it: a rank that raised here while the others entered the step's all-reduce would hang the job until the
    # process group's time-out and cost the headline line.
    setup_error = None
    try:
        same, check_replays = replay_equals_eager(args, rank, device, batch, layers, model_kind)
        params, compute, store, what = build_train_workload(args, rank, device, batch, layers, model_kind)
        arena = D.GradArena(params)              # every gradient of the model in ONE flat buffer = the one message of the step
        opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
    except Exception as e:       # noqa: BLE001
        setup_error = e
    if dist is not None:
        ok = torch.tensor([0.0 if setup_error is not None else 1.0], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and setup_error is None:
            setup_error = RuntimeError("another rank failed to set the training leg up")
    if setup_error is not None:
        raise setup_error
    ev = []
    counted = []

    nstep = [0]
    if Watchdog._one is not None:
        Watchdog._one.lib_handle = _lib_handle()          # (the training engine's host-side counters, for a stall's dump)

    def step(record=False):
        nstep[0] += 1
        Watchdog.beat("train step %d: forward / backward" % nstep[0])
        arena.zero_grad()                                            # training.py:54 (one fill)
        loss = compute()                                             # training.py:37,42
        loss.backward()                                              # training.py:55
        Watchdog.beat("train step %d: the gradient all-reduce" % nstep[0])
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if dist is not None:
            with D.count_collectives() as cc:
                n = arena.allreduce()                                # the ONE collective of the step
            counted.append(cc.total())
        else:
            n = arena.allreduce()
        if record:
            e1.record()
            ev.append((e0, e1))
        opt.step()                                                   # training.py:56
        return loss, n

    for _ in range(warmup):
        loss, nred = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, nred = step(record=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ar_all = sorted(float(a.elapsed_time(b)) * 1e3 for a, b in ev)
    ar_us = float(np.median(ar_all))
    nbytes = arena.nbytes()
    gstats = (ctypes.c_long * 5)()
    _lib_handle().dpf_train_graph_stats(gstats)          # of the timed region: read before the eager per-kernel pass below
    _lib_handle().dpf_train_colsum_fallbacks.restype = ctypes.c_long
    colsum_fallbacks = int(_lib_handle().dpf_train_colsum_fallbacks())   # (workgroups that gave up waiting for role workgroups)
    # ---- per-kernel durations of the decoder stack (outside the timed region): HIP events around every launch, eager
    kernels = None
    try:
        L_ = _lib_handle()
        ksteps = 3
        us, calls = (ctypes.c_double * 8)(), (ctypes.c_long * 8)()
        L_.dpf_train_kernel_times(1, None, None)        # turns graph recording / replay off process-wide ...
        try:
            for _ in range(ksteps):
                step()
        finally:
            L_.dpf_train_kernel_times(0, us, calls)     # ... and only this call restores it: also when a step raises
        P = batch * N
        # FLOPs a kernel executes per layer for P points (both branches): conditioner forward 2 x (2*F*|K| + 2*F*F) ~ 16 896 / pt
        # (|K| ~ 1.5 on average), output layer 2 x 2*|W|*F ~ 384, W1^T dh1 and dh1 h0^T 2 x 2*F*F = 16 384 each
        fwd, out_l, sq = 16896.0 * P, 384.0 * P, 16384.0 * P
        flop = {"tstats_h1": fwd, "flow_kernel(L=1)": fwd + out_l, "tbwd1": fwd + out_l, "tbwd2": fwd + 2 * sq, "tfold": 0.0, "tcolsum": 0.0,
                "tstats_x": 0.0, "tbwd3f": 0.0}
        names = ["tstats_x", "tstats_h1", "tfold", "flow_kernel(L=1)", "tbwd1", "tbwd2", "tcolsum", "tbwd3f"]
        kernels, per_layer = {}, 0.0
        for i, nm in enumerate(names):
            if calls[i] == 0:
                continue
            t_us = us[i] / calls[i]
            per = us[i] / (ksteps * layers)
            per_layer += per
            kernels[nm] = {"us": t_us, "launches_per_step": calls[i] / ksteps, "us_per_layer": per,
                           "gflop": flop[nm] / 1e9, "tflops": (flop[nm] / (t_us * 1e-6) / 1e12) if flop[nm] else None,
                           "frac_of_mfma_peak": (flop[nm] / (t_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF) if flop[nm] else None}
        kernels["sum_us_per_layer"] = per_layer
        kernels["note"] = ("HIP events around every launch of the stack's kernels, %d eager steps (dpf_train_kernel_times): event to "
                           "event, i.e. kernel duration + the ~1.5 us launch gap of an eager launch -- the rocprofv3 kernel durations of "
                           "the same command are in profiles/r06_train_kernel_trace.txt; gflop = "
                           "what the kernel executes per launch incl. the recomputation of the conditioner; tfold / tcolsum / "
                           "tstats_x / tbwd3f are reductions (latency-bound, no matrix work)" % ksteps)
    except Exception as e:       # noqa: BLE001
        kernels = {"error": repr(e)}
    info = {"ms_per_step": elapsed / steps * 1e3, "value": batch * world * N / (elapsed / steps) if same else None, "unit": "points/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "layers": layers, "clouds_per_gpu": batch, "points_per_cloud": N,
            "latent": G, "loss": float(loss.detach()), "precision": _train_precision(),
            "model": model_kind,
            "what": "zero_grad + " + what + " + backward + ONE all-reduce of the model's flat gradient + Adam (AMSGrad mirror)",
            "encoder": getattr(args, "encoder", "none"),
            "parameters": int(arena.n_params), "flat_gradient_bytes": nbytes,
            "collectives_per_step": (max(counted) if counted else 0),
            "collectives_counted": bool(counted),
            "replay_equals_eager": bool(same), "replay_check": "first 6 optimizer steps, hipGraph replay vs eager launches, loss "
                                                               "bytes compared (%d calls of the check were replays)" % check_replays,
            "colsum_fallbacks": colsum_fallbacks,
            "graph_replays": int(gstats[0]), "graph_stats": {"replays": int(gstats[0]), "eager_calls": int(gstats[1]),
                                                             "recordings": int(gstats[2]), "evictions": int(gstats[3]),
                                                             "uncapturable": int(gstats[4])},
            "algorithmic_tflops": 3.0 * FLOP_PER_POINT_LAYER * layers * batch * N / (elapsed / steps) / 1e12,
            "kernels": kernels}
    if not same:
        info["error"] = "graph replay and eager launches disagree: no value reported"
    if world > 1:
        bus = 2.0 * (world - 1) / world * nbytes / (ar_us * 1e-6) / 1e9
        info["allreduce"] = {"us": ar_us, "us_min": ar_all[0], "us_max": ar_all[-1], "us_p10": ar_all[len(ar_all) // 10],
                             "us_p90": ar_all[(9 * len(ar_all)) // 10], "steps_timed": len(ar_all),
                             "frac_of_step": ar_us * 1e-3 / (elapsed / steps * 1e3),
                             "rccl": rccl_log_excerpt() if dist.get_backend() == "nccl" else None,
                             "elements": int(nred), "bus_GBps": bus, "backend": dist.get_backend(),
                             "xgmi_budget_GBps": 7 * XGMI_LINK_GBS, "frac_of_xgmi_budget": bus / (7 * XGMI_LINK_GBS),
                             "note": "HIP events on the compute stream around dist.all_reduce(arena.buf) (includes the wait for the "
                                     "collective's stream); bus = 2(n-1)/n * bytes / time"}
    return info


def leg_train(args, rank, world, dist, device):
    batch, global_clouds = clouds_of_rank(args, rank, world)
    info = train_step_leg(args, rank, world, dist, device, batch, args.layers, args.steps, args.warmup)
    if rank != 0:
        return None, {}
    cfg = CONFIGS[args.config]
    tf = info["algorithmic_tflops"]
    line = {"metric": "points/sec through a training step of the %d-layer flow decoder, B=%d N=%d" % (args.layers, batch, args.points),
            "value": info["value"], "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": info["ms_per_step"], "higher_is_better": True,
            "scaling": ("strong" if getattr(args, "strong", False) else cfg["scaling"]) if args.batch is None else "weak",
            "vs_baseline": None, "dtype": info["precision"] + " MFMA operands, fp32 accumulate/points/gradients",
            "data": "synthetic",
            "config": {"workload": cfg["name"] + ": " + info["what"], "clouds_per_gpu": batch, "points_per_cloud": args.points,
                       "global_clouds": batch * world, "layers": args.layers, "latent": args.latent,
                       "launch": "eager Python step; the stack's ~700 kernel launches replay as hipGraphs (csrc/graph_cache.h, "
                                 "DPF_TRAIN_GRAPH=0 to switch off): %d graph launches so far" % info.get("graph_replays", 0),
                       "parallelism": "data parallel replicas, one all-reduce of the flat gradient per step"},
            "roofline": {"kernel": "training step (tbwd2/tbwd1/tstats_h1/flow kernels, csrc/flow_train.hip)", "bound": "mfma",
                         "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                         "traffic": None, "note": "whole step incl. host: 3 x forward FLOPs (forward + two backward contractions) "
                                                  "/ step time",
                         "kernels": info.get("kernels")},
            "train_step": info, "cpu_baseline": None, "parity": None}
    return line, {}


def extra_config_legs(args, rank, world, device):
    """Short legs of the OTHER workloads SURVEY 8(d) names, measured in the same process as the headline (rank-local, no
    collective: the eval path shards by cloud): L = 63 and L = 15 at cfg2, cfg3 (all-classes shapes), cfg4 (SVR shapes +
    f_score), cfg5 (dense Chamfer + approx-EMD).  Each with ms/step, value, roofline fraction and parity; ~1.5 s per leg."""
    import copy
    out = {}
    legs = [("cfg2_L63", dict(config="cfg2", layers=63)), ("cfg2_L15", dict(config="cfg2", layers=15)),
            ("cfg3_L14", dict(config="cfg3", layers=14)), ("cfg4_L14", dict(config="cfg4", layers=14)),
            ("cfg5", dict(config="cfg5"))]
    for name, over in legs:
        try:
            a2 = copy.copy(args)
            cfg = CONFIGS[over["config"]]
            a2.config, a2.batch, a2.points, a2.latent = over["config"], None, cfg["points"], cfg["latent"]
            a2.layers = over.get("layers", 14)
            a2.no_extra, a2.no_cpu_baseline, a2.pipelined, a2.streams = True, True, 0, 1
            if over["config"] == "cfg5":
                a2.steps, a2.warmup, a2.settle = 4, 2, 0
                line, _ = leg_cfg5(a2, rank, world, None, device)
            else:
                a2.steps, a2.warmup, a2.settle = 200, 50, 150
                line, _ = leg_eval(a2, rank, world, None, device)
            if line is None:
                continue
            r = line["roofline"]
            out[name] = {"workload": line["config"]["workload"], "value": line["value"], "unit": line["unit"],
                         "ms_per_step": line["ms_per_step"], "steps": a2.steps, "warmup": a2.warmup,
                         "clouds_per_gpu": line["config"]["clouds_per_gpu"], "points_per_cloud": line["config"]["points_per_cloud"],
                         "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "kernels_us", "hbm")
                                      if r.get(k) is not None},
                         "parity": line.get("parity")}
        except Exception as e:       # noqa: BLE001 -- an extra must never cost the headline line
            out[name + "_error"] = repr(e)
        torch.cuda.empty_cache()
    return out


def per_rank_proxy(args, device):
    """what ONE rank of an 8-GPU STRONG-scaling run does, measured on this one GPU (the pool has no 8-GPU
    node for the builder; the driver's SCALE run supplies the real curve): the eval step (FiLM + fused stack + Chamfer + CD)
    at the 4 clouds a rank holds of BASELINE's B = 32 and at 8 clouds, and the 63-layer training step at the 8 clouds (G = 512) a
    rank holds of configs[2]'s B = 64 (configs/autoencoding/all_scaled.yaml:5,9).  Projection for 8 GPUs = B_global * N /
    (t(B_local) + all-reduce): the eval path has no collective; the training step's all-reduce of the flat gradient is
    MODELLED (not measured here) two ways from SURVEY 8(e)'s link budget -- a ring bound by one 153 GB/s xGMI link, and a
    direct reduce-scatter + all-gather over all 7 links."""
    import copy
    out = {"what": "1-GPU stand-ins for one rank of an 8-GPU strong-scaling job; projections, not measurements of 8 GPUs",
           "eval": {}, "train": {}}
    for bl in (4, 8):
        try:
            a2 = copy.copy(args)
            a2.batch, a2.strong = bl, False
            a2.no_extra, a2.no_cpu_baseline, a2.pipelined, a2.streams = True, True, 0, 1
            a2.steps, a2.warmup, a2.settle = 400, 100, 200
            line, _ = leg_eval(a2, 0, 1, None, device)
            t = line["ms_per_step"] * 1e-3
            out["eval"]["B_local_%d" % bl] = {
                "ms_per_step": line["ms_per_step"], "points_per_s_this_gpu": line["value"],
                "kernels_us": line["roofline"].get("kernels_us"),
                "flow_kernel": line["config"].get("flow_kernel"),
                "projected_8gpu_points_per_s": 8 * bl * args.points / t,
                "global_batch": 8 * bl, "collectives": 0, "parity": line.get("parity")}
        except Exception as e:       # noqa: BLE001 -- an extra must never cost the headline line
            out["eval"]["B_local_%d_error" % bl] = repr(e)
        torch.cuda.empty_cache()
    try:
        a3 = copy.copy(args)
        a3.latent, a3.batch, a3.strong = 512, 8, False
        info = train_step_leg(a3, 0, 1, None, device, 8, 63, 24, 12)
        t = info["ms_per_step"] * 1e-3
        nbytes = info["flat_gradient_bytes"]
        ring = 2.0 * 7 / 8 * nbytes / (XGMI_LINK_GBS * 1e9)
        direct = 2.0 * (nbytes / 8) / (XGMI_LINK_GBS * 1e9)
        out["train"]["B_local_8_G512_L63"] = {
            "ms_per_step": info["ms_per_step"], "points_per_s_this_gpu": info["value"], "flat_gradient_bytes": nbytes,
            "replay_equals_eager": info.get("replay_equals_eager"),
            "kernels_us_per_layer": {k: v.get("us_per_layer") for k, v in (info.get("kernels") or {}).items() if isinstance(v, dict)},
            "allreduce_model_ms": {"ring_one_link": ring * 1e3, "all_links": direct * 1e3},
            "allreduce": {"range_ms": [direct * 1e3, ring * 1e3],
                          "source": "MODELLED, never measured on RCCL (no multi-GPU node in the builder's pool): low = reduce-scatter + all-gather over "
                                    "all 7 xGMI links at %.0f GB/s each, high = a ring bound by one link; the whole term is the projection's "
                                    "error bar" % XGMI_LINK_GBS,
                          "functional_only": "two ranks SHARING one GPU over gloo run the same code path end to end "
                                             "(profiles/r04_bench_2ranks_sharing_one_gpu_gloo.json, tests/test_gpu_multi.py): correctness of the "
                                             "one-collective step, no statement about RCCL / xGMI time",
                          "measured_when_available": "bench.py --leg train --gpus N > 1 reports extra-free: allreduce.us (median) / us_min / "
                                                     "us_p10 / us_p90 / us_max per step, frac_of_step, bus_GBps and RCCL's own INIT / GRAPH "
                                                     "lines (allreduce.rccl)"},
            "projected_8gpu_points_per_s": {"ring_one_link": 64 * args.points / (t + ring), "all_links": 64 * args.points / (t + direct)},
            "global_batch": 64, "collectives": 1}
    except Exception as e:       # noqa: BLE001
        out["train"]["B_local_8_G512_L63_error"] = repr(e)
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------------------------------
def selftest_ranks(args, rank, world, dist):
    """DPF_BENCH_SELFTEST=1: what the CPU test of the launcher runs instead of the GPU legs -- proves that `--gpus N`
    started N ranks which see each other (gloo), and that the max-over-ranks reduction and the flat all-reduce work."""
    t = torch.tensor([float(rank + 1)])
    flat = torch.arange(8, dtype=torch.float32) * (rank + 1)
    Watchdog.beat("selftest: before the first all-reduce")
    if os.environ.get("DPF_BENCH_SELFTEST_STALL_RANK") == str(rank):      # (the watchdog's own test: this rank never arrives)
        time.sleep(3600)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(flat)
        dist.barrier()
    if rank == 0:
        line = {"selftest": True, "n_gpus": world, "max_rank_plus_1": float(t.item()),
                "flat_sum_ok": bool(torch.equal(flat, torch.arange(8, dtype=torch.float32) * (world * (world + 1) / 2)))}
        if os.environ.get("DPF_BENCH_SELFTEST_STALL_AFTER_LINE") == "1":      # (the watchdog's own test: an `extra` that never returns)
            Watchdog.keep(line)
            Watchdog.beat("selftest: an extra leg behind the measured line")
            time.sleep(3600)
        print(json.dumps(line))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    rank, local_rank, world, dist = init_ranks(args)
    wd = Watchdog.start(rank)
    if os.environ.get("DPF_BENCH_SELFTEST") == "1":
        selftest_ranks(args, rank, world, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if os.environ.get("DPF_BENCH_SHARE_GPU") == "1":      # test mode: the ranks share cuda:0 (with DPF_BENCH_BACKEND=gloo; RCCL wants one device per rank)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    Watchdog.beat("leg %s / %s" % (args.leg, args.config))
    if args.leg == "train":
        line, extra = leg_train(args, rank, world, dist, device)
    elif args.config == "cfg5":
        line, extra = leg_cfg5(args, rank, world, dist, device)
    else:
        line, extra = leg_eval(args, rank, world, dist, device)
        if rank == 0:
            Watchdog.keep(dict(line, extra=dict(extra)))      # (a stall in one of the extras below prints this, see Watchdog._watch)
        if not args.no_extra and args.config == "cfg2" and args.layers == 14 and not args.no_configs:
            try:
                Watchdog.beat("extra: the other configs")
                extra["configs"] = extra_config_legs(args, rank, world, device)
            except Exception as e:       # noqa: BLE001
                extra["configs_error"] = repr(e)
        if not args.no_extra:
            # the training step with its single gradient all-reduce, on the default run too: at N > 1 this is where RCCL carries
            # the 40 / 52 MB flat gradient over xGMI (every rank takes part; reported by rank 0)
            try:
                Watchdog.beat("extra: training step leg")
                batch, _ = clouds_of_rank(args, rank, world)
                extra["train_step"] = train_step_leg(args, rank, world, dist, device, batch, 63, args.train_steps, 16)
            except Exception as e:       # noqa: BLE001 -- never lose the headline line to an extra
                extra["train_step_error"] = repr(e)
        if not args.no_extra and not args.no_proxy and world == 1 and args.config == "cfg2" and args.layers == 14 and args.batch is None:
            try:                                         # rank-local, no collective: the single-GPU run only
                Watchdog.beat("extra: per-rank proxies")
                extra["per_rank_proxy"] = per_rank_proxy(args, device)
            except Exception as e:       # noqa: BLE001
                extra["per_rank_proxy_error"] = repr(e)
    if rank == 0:
        Watchdog.keep(None)
        if extra:
            line["extra"] = extra
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
