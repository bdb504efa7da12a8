#!/bin/bash
# Kernel trace of the training leg for several builds of the library (same box): tools/train_ab_prof.sh <tag> <so> [<so> ...]
# prints per build the summed average kernel time of the per-layer kernels (us per layer)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$so -o kt -- python3 $GRAFT_REPO_ROOT/tools/lib_ab.py $so --leg train --steps 12 --warmup 4 > $OUT/kt_$so.log 2>&1
  for db in $(find $OUT/kt_$so -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/trace_$so.txt 2>&1; done
  rm -rf $OUT/kt_$so
  python3 - "$OUT/trace_$so.txt" "$so" <<'PY'
import sys
tot = 0.0; rows = []; base = None
names = ("tbwd2_kernel", "tbwd1_kernel", "tstats_h1_kernel", "flow_kernel", "tfold_kernel", "tcolsum_kernel", "tbwd3f", "tstats_x", "dpf_zero_words")
found = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) > 4 and p[1].isdigit():
        for n in names:
            if n in p[0]:
                found[n] = (int(p[1]), float(p[2]))
base = found["tbwd2_kernel"][0]
for n in names:
    if n in found:
        calls, avg = found[n]
        per = calls / base * avg / 1e3
        tot += per
        rows.append("%s %.2f" % (n.replace("_kernel", ""), per))
print(sys.argv[2], "per-layer kernel sum %.1f us |" % tot, "; ".join(rows))
PY
done
