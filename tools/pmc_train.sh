#!/bin/bash
# Two PMC passes over the training leg (eager launches; counters in their own runs): tools/pmc_train.sh <outdir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DPF_TRAIN_GRAPH=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --leg train --steps 4 --warmup 2 "$@" > $OUT/pass$i.log 2>&1
  for db in $(find $OUT/pass$i -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/pass$i.txt 2>&1; done
  rm -rf $OUT/pass$i
done
grep -h "tbwd\|tstats_h1\|flow_kernel" $OUT/pass1.txt $OUT/pass2.txt | cut -c15-40,60-140 | head -80
