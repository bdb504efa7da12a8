"""N consecutive two-rank runs of the training leg, the ranks sharing this box's one GPU over gloo (functional only),
each under bench.py's watchdog: how many complete, and for every run that does not, the watchdog's line (which rank, where it stood,
the training engine's counters).    python tools/two_rank_soak.py [N] [out.json]"""
import json, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
env = dict(os.environ, DPF_BENCH_BACKEND="gloo", DPF_BENCH_SHARE_GPU="1", DPF_BENCH_WATCHDOG_S="90", DPF_BENCH_LAUNCH_TIMEOUT_S="400")
rows = []
for i in range(N):
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--leg", "train", "--steps", "8", "--warmup", "4"],
                       capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    row = {"run": i, "rc": r.returncode, "seconds": round(time.time() - t0, 1)}
    if r.returncode == 0 and line:
        d = json.loads(line[-1])
        row.update(ms_per_step=d["ms_per_step"], collectives_per_step=d.get("extra", {}).get("collectives_per_step", d.get("collectives_per_step")))
    else:
        row["watchdog"] = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{") and ("watchdog" in l or "launcher" in l)]
        row["stderr_tail"] = r.stderr[-600:]
    rows.append(row)
    print(json.dumps(row), flush=True)
out = {"command": "DPF_BENCH_BACKEND=gloo DPF_BENCH_SHARE_GPU=1 DPF_BENCH_WATCHDOG_S=90 python bench.py --gpus 2 --leg train --steps 8 --warmup 4",
       "runs": N, "completed": sum(1 for r in rows if r["rc"] == 0), "rows": rows}
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(json.dumps(out, indent=1))
print(json.dumps({k: out[k] for k in ("runs", "completed")}))
