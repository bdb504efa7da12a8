#!/bin/bash
# Kernel trace of the training leg (decoder step) + the bench lines of the training variants: gpurun_out/<tag>/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --leg train --steps 12 --warmup 4 > $OUT/kt.log 2>&1
for db in $(find $OUT/kt -name "*.db"); do
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --leg train --steps 12 --warmup 4"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db; } > $OUT/train_kernel_trace.txt 2>&1
done
rm -rf $OUT/kt
cd $GRAFT_REPO_ROOT
python3 bench.py --leg train --steps 40 --warmup 10 > $OUT/bench_train.json 2> $OUT/err.log
for e in none tensor hip; do python3 bench.py --leg train --encoder $e --steps 40 --warmup 10 > $OUT/bench_train_encoder_$e.json 2>> $OUT/err.log; done
DPF_TRAIN_GRAPH=0 python3 bench.py --leg train --steps 40 --warmup 10 > $OUT/bench_train_nograph.json 2>> $OUT/err.log
ls $OUT
