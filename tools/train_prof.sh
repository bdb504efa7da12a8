#!/bin/bash
# Kernel trace of the training leg (decoder step) + the bench lines of the training variants: gpurun_out/<tag>/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --leg train --steps 12 --warmup 4 > $OUT/kt.log 2>&1
for db in $(find $OUT/kt -name "*.db"); do
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --leg train --steps 12 --warmup 4   (28 optimizer steps in all: 12 of the replay-vs-eager check, 4 + 12 of the leg; + 3 eager steps of the per-kernel event timing)"; echo "# the ~10 000 __amd_rocclr_copyBuffer calls below are NOT part of a training step: they are the replay-vs-eager check's saves and restores of the model / optimizer state (bench.py::replay_equals_eager, 12 steps x 2 runs), identical in count since r03"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db; } > $OUT/train_kernel_trace.txt 2>&1
done
rm -rf $OUT/kt
cd $GRAFT_REPO_ROOT
python3 bench.py --leg train --steps 40 --warmup 12 > $OUT/bench_train.json 2> $OUT/err.log
python3 bench.py --leg train --model autoencoder --steps 40 --warmup 12 > $OUT/bench_train_autoencoder_cfg2.json 2>> $OUT/err.log
python3 bench.py --leg train --config cfg3 --model autoencoder --steps 40 --warmup 12 > $OUT/bench_train_autoencoder_cfg3.json 2>> $OUT/err.log
DPF_TRAIN_GRAPH=0 python3 bench.py --leg train --steps 40 --warmup 12 > $OUT/bench_train_nograph.json 2>> $OUT/err.log
DPF_TRAIN_PRECISION=bf16x6 python3 bench.py --leg train --steps 40 --warmup 12 > $OUT/bench_train_bf16x6.json 2>> $OUT/err.log
python3 tests/diag/replay_vs_eager.py 50 21 > $OUT/replay_vs_eager.txt 2>&1
ls $OUT
