#!/bin/bash
# Where do the __amd_rocclr_copyBuffer launches of the training step come from?  kernel + HIP API trace of the train leg with
# graph replay on and off: gpurun_out/<tag>/{graph,eager}_{kernels,hipapi}.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in graph eager; do
  if [ $mode = eager ]; then export DPF_TRAIN_GRAPH=0; else export DPF_TRAIN_GRAPH=1; fi
  rocprofv3 --kernel-trace --hip-trace --stats -d $OUT/$mode -o t -- python3 $GRAFT_REPO_ROOT/bench.py --leg train --steps 8 --warmup 8 > $OUT/$mode.log 2>&1
  for db in $(find $OUT/$mode -name "*.db"); do
    python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db | head -12 > $OUT/${mode}_kernels.txt
    python3 - $db > $OUT/${mode}_hipapi.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
print([t for t in tabs if 'region' in t or 'api' in t or 'string' in t][:20])
try:
    q = """select s.string, count(*), avg(r.end-r.start) from rocpd_region r join rocpd_string s on r.name_id = s.id group by s.string order by 2 desc limit 25"""
    for row in cur.execute(q):
        print("%-48s %8d %10.0f" % row)
except Exception as e:
    print("query failed", e)
PY
  done
  rm -rf $OUT/$mode
done
unset DPF_TRAIN_GRAPH
head -30 $OUT/*_kernels.txt $OUT/*_hipapi.txt
