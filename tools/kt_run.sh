#!/bin/bash
# rocprofv3 kernel trace of one bench.py command line (run on the GPU box through gpurun):
#   tools/kt_run.sh <out-name> <bench.py arguments...>   -> gpurun_out/<out-name>.txt
set -u
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/kt_$NAME
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_$NAME -o kt -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/kt_$NAME.log 2>&1
for db in $(find $OUT/kt_$NAME -name "*.db"); do
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $*"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db; } > $OUT/$NAME.txt 2>&1
done
rm -rf $OUT/kt_$NAME
