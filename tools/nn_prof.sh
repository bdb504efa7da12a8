cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof7 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof7 -name "*.db") | head -7 | cut -c1-150
