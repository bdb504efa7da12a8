#!/bin/bash
# PMC passes over the training-mode encoder (counters in their own runs, no trace domains combined).
# usage: tools/encoder_train_pmc.sh <outdir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/tools/encoder_train_bench.py 32 2048 2 hiponly > $OUT/pass$i.log 2>&1
  for db in $(find $OUT/pass$i -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/pass$i.txt 2>&1; done
  rm -rf $OUT/pass$i
done
