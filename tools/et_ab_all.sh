#!/bin/bash
# per-kernel times for each ablated build: gpurun_out/etab/<mask>.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/etab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  so=libdpf_et$m.so; [ "$m" = "0" ] && so=libdpf_hip.so
  rocprofv3 --kernel-trace --stats -d $OUT/kt$m -o kt -- python3 $GRAFT_REPO_ROOT/tools/et_ab_run.py $so > $OUT/$m.log 2>&1
  for db in $(find $OUT/kt$m -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db | grep -E "pgemm" | awk -v m=$m '{print m, $1, $3}' | sed 's/_ZN12_GLOBAL__N_115et_pgemm_kernelI//;s/EEvNS_5PArgs.*\s/ /' > $OUT/$m.txt; done
  rm -rf $OUT/kt$m
done
paste $OUT/*.txt | head -20
