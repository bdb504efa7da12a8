"""Timings of the hot path at every BASELINE.json configuration shape (per-GPU slice)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [
    ("cfg-1 single layer B=4 N=256", ["--batch", "4", "--points", "256", "--layers", "1"]),
    ("cfg-2 L=14 B=32 N=2048 G=128", []),
    ("cfg-2 L=15 (n_flows=5)", ["--layers", "15"]),
    ("cfg-2 L=63 (n_flows=21, shipped config)", ["--layers", "63"]),
    ("cfg-2 L=63 with per-layer lists", ["--layers", "63", "--lists"]),
    ("cfg-3 per-GPU slice B=8 N=2048 G=512 L=63", ["--batch", "8", "--latent", "512", "--layers", "63"]),
    ("cfg-4 SVR decoder B=32 N=2500 G=512 L=63", ["--batch", "32", "--points", "2500", "--latent", "512", "--layers", "63"]),
    ("cfg-5 per-GPU slice B=2 N=8192 L=63", ["--batch", "2", "--points", "8192", "--layers", "63"]),
    ("cfg-5 whole B=16 N=8192 L=63 on one GPU", ["--batch", "16", "--points", "8192", "--layers", "63"]),
]
for name, extra in CONFIGS:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "50",
                          "--warmup", "10"] + extra, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = d["roofline"]["kernels_us"]
        print("%-46s %9.3e pts/s  step %8.1f us | film %6.1f flow %8.1f nn %8.1f us" % (
            name, d["value"], d["ms_per_step"] * 1e3, k["film_kernel"], k["flow_kernel"], k["nn_kernel"]), flush=True)
    except Exception as ex:
        print(name, "FAILED", ex, out.stderr[-400:])
