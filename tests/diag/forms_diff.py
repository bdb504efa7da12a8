"""Diagnostic: run the training step in each kernel form (child processes) and list the largest differences from the default
form, outputs included -- for telling a rounding difference from a differing ReLU mask.  Usage: python tests/diag/forms_diff.py"""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_flow_train as T
code = (T.FORMS_CODE % (root, T.FORMS_SHAPES)).replace('    key = "%d_%d" % (B, N)\n', '    key = "%d_%d" % (B, N)\n    for i, p in enumerate(ps): out[key + "/out%d" % i] = p.detach().cpu().numpy()\n')
forms = {"default": {}, "pair": {"DPF_TRAIN_SPLIT": "0"}, "split": {"DPF_TRAIN_SPLIT": "1"}, "ticket": {"DPF_TRAIN_ROLES": "0"}}
res = {}
d = tempfile.mkdtemp()
for name, env in forms.items():
    f = os.path.join(d, name + ".npz")
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (name, r.stderr[-2000:])
    res[name] = dict(np.load(f))
base = res["default"]
for name, got in res.items():
    if name == "default":
        continue
    rows = []
    for k in base:
        if k == "fallbacks":
            continue
        s = float(np.abs(base[k]).max())
        if s == 0:
            continue
        rows.append((float(np.abs(got[k].astype(np.float64) - base[k]).max()) / s, k))
    rows.sort(reverse=True)
    print(name, " ".join("%s=%.1e" % (k, e) for e, k in rows[:8]))
    print(name, "outputs", " ".join("%s=%.1e" % (k, e) for e, k in rows if "/out" in k and e > 0))
for k in base:
    if k.endswith("mu_sd2.bias") and k.startswith("8_2048"):
        w = k[:-4] + "weight"
        print(k, "bias", base[k], "pair", res["pair"][k], "|w|max", float(np.abs(base[w]).max()))
