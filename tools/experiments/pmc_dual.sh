#!/bin/bash
# two PMC passes over bench.py with the DUAL flow body (DPF_FLOW_DUAL=1): tools/pmc_dual.sh <outdir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DPF_FLOW_DUAL=${DPF_FLOW_DUAL:-1}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pass$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --pipelined 0 --no-graph --settle 0 --steps 20 --warmup 5 "$@" > $OUT/pass$i.log 2>&1
  for db in $(find $OUT/pass$i -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $OUT/pass$i.txt 2>&1; done
  rm -rf $OUT/pass$i
done
grep -h "flow_" $OUT/pass1.txt $OUT/pass2.txt | cut -c1-20,60-140
