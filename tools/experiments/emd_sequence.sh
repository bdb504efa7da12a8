cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kts -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > /tmp/kts.log 2>&1
for db in $(find /tmp/kts -name "*.db"); do python3 $GRAFT_REPO_ROOT/tools/rocprof_sequence.py $db emd_ 60 48; done
