"""profiles/traffic.json entry "approxmatch_cost/B<b>_N<n>": HBM bytes of ONE ApproxMatchCost call (all of its emd_* launches) from
the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_cfg5.sh, corrected as MI355X_MICROARCH.md prescribes (both counters in KiB;
FETCH_SIZE tallies 128-B requests at 64 B on gfx950 -> x2).   usage: pmc_traffic_emd.py <pmc dir> <B> <N> <traffic.json> [source note]"""
import json
import os
import re
import sys


def parse(path):
    calls, vals = {}, {}
    for line in open(path):
        p = line.split()
        if len(p) > 10 and p[1].isdigit() and "emd_" in p[0]:
            calls[p[0]] = int(p[1])
        m = re.match(r"(\S+)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
        if m and "emd_" in m.group(1):
            vals[(m.group(1), m.group(2))] = (float(m.group(3)), int(m.group(4)))
    return calls, vals


def main(d, B, N, out_json, source="profiles/r03_pmc_cfg5_pass1.txt + _pass2.txt"):
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    ncalls = None
    per_kernel = {}
    for f in sorted(os.listdir(d)):
        if not f.endswith(".txt"):
            continue
        calls, vals = parse(os.path.join(d, f))
        for (k, c), (avg, n) in vals.items():
            # the per-dispatch average is over counter INSTANCES (n = dispatches x instances): avg x dispatches x instances / dispatches
            disp = calls.get(k, 0)
            inst = n // max(disp, 1)
            tot[c] += avg * inst * disp
            per_kernel.setdefault(k[:48], {})[c] = avg * inst
            if "cost" in k or "mat" in k:
                ncalls = disp if ncalls is None else min(ncalls, disp)
    ncalls = ncalls or 1
    e = {"fetch_bytes_raw": tot["FETCH_SIZE"] * 1024 / ncalls, "fetch_bytes": tot["FETCH_SIZE"] * 1024 * 2 / ncalls,
         "write_bytes": tot["WRITE_SIZE"] * 1024 / ncalls, "calls": ncalls, "source": source,
         "per_kernel_KiB_per_dispatch": per_kernel}
    e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    res = json.load(open(out_json)) if os.path.exists(out_json) else {}
    res["approxmatch_cost/B%s_N%s" % (B, N)] = e
    json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps(e, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(*sys.argv[1:6])
