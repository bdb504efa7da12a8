"""The driver's exact command, R times in fresh processes on THIS box: min / median / max of the line's ms_per_step and of the flat
scalars that tell a slow box from a regression from a gap in the window.
    python tools/driver_repeat.py [R] [out.json]"""
import json, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = []
for i in range(R):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        rows.append({"rc": r.returncode, "stderr": r.stderr[-400:]})
        continue
    d = json.loads(line[-1])
    rows.append({"us_per_step": d["ms_per_step"] * 1e3, "us_per_step_wall": d["config"]["ms_per_step_wall"] * 1e3,
                 "kernels_sum_us": d["roofline"]["kernels_sum_us"], "flow_us": d["roofline"]["flow_us"], "nn_us": d["roofline"]["nn_us"],
                 "film_us": d["roofline"]["film_us"], "sclk_mhz": d["config"]["sclk_mhz"], "value": d["value"],
                 "train_ms": d.get("extra", {}).get("train_step", {}).get("ms_per_step")})
ok = [r for r in rows if "us_per_step" in r]
def mmm(k):
    v = np.array([r[k] for r in ok if r.get(k) is not None], float)
    return {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max())} if len(v) else None
out = {"command": "python3 bench.py --gpus 1 --steps 20 --warmup 5", "runs": R, "ok": len(ok),
       "summary": {k: mmm(k) for k in ("us_per_step", "us_per_step_wall", "kernels_sum_us", "flow_us", "nn_us", "film_us", "sclk_mhz", "train_ms")},
       "rows": rows}
s = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s)
print(json.dumps(out["summary"]))
