#!/bin/bash
# HBM counter passes (FETCH_SIZE, WRITE_SIZE -- each in its own run, no trace domains) over the cfg5 leg (Chamfer + approx-EMD,
# B=16 N=M=8192): gpurun_out/<tag>/pass{1,2}.txt ; tools/pmc_traffic_emd.py turns them into profiles/traffic.json entries.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/pass$i.log 2>&1
  for db in $(find $OUT/pass$i -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/pass$i.txt 2>&1; done
  rm -rf $OUT/pass$i
done
