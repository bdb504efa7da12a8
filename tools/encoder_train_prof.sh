#!/bin/bash
# Kernel trace of the training-mode encoder (forward + backward, cfg-2 shape): gpurun_out/<tag>/enc_train_kernels.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/encoder_train_bench.py ${2:-32} ${3:-2048} 20 > $OUT/kt.log 2>&1
for db in $(find $OUT/kt -name "*.db"); do
  { echo "# rocprofv3 --kernel-trace --stats -- python3 tools/encoder_train_bench.py ${2:-32} ${3:-2048} 20"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db; } > $OUT/enc_train_kernels.txt 2>&1
done
rm -rf $OUT/kt
tail -3 $OUT/kt.log
