#!/bin/bash
# Kernel trace of the autoencoder training leg: tools/ae_trace.sh <tag> [bench args]  -> gpurun_out/<tag>/ae_trace.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --leg train --model autoencoder --steps 12 --warmup 4 "$@" > $OUT/kt.log 2>&1
for db in $(find $OUT/kt -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/ae_trace.txt 2>&1; done
rm -rf $OUT/kt
python3 - "$OUT/ae_trace.txt" <<'PY'
import sys
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) > 4 and p[1].isdigit():
        rows.append((p[0], int(p[1]), float(p[2])))
steps = [c for n, c, a in rows if "adam_kernel" in n][0]
tot = sum(c * a for n, c, a in rows if "copyBuffer" not in n) / steps / 1e3
print("steps (adam launches)", steps, " kernel time per step %.1f us" % tot)
for n, c, a in sorted(rows, key=lambda r: -r[1] * r[2])[:40]:
    if "copyBuffer" in n: continue
    print("%-86s %6.1f calls/step %8.1f us/step" % (n[:86], c / steps, c * a / steps / 1e3))
PY
