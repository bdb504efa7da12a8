#!/bin/bash
# One call that collects everything profiles/ needs for a round (run on the GPU box through gpurun):
#   tools/profile_round.sh r02        -> gpurun_out/r02_prof/{kernel_trace_stats.txt, pmc/pass*.txt, traffic.json, bench_*.json}
# Kernel trace and counters come from SEPARATE rocprofv3 runs (never combined with trace domains); the profiled program is
# python3 bench.py itself, directly after `--`.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --pipelined 0 > $OUT/kt.log 2>&1
for db in $(find $OUT/kt -name "*.db"); do
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extra --pipelined 0   ($TAG; hipGraph replay, default precision)"; python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db; } > $OUT/kernel_trace_stats.txt 2>&1
done
rm -rf $OUT/kt
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh ${TAG}_prof/pmc > /dev/null 2>&1
cp profiles/traffic.json $OUT/traffic.json 2>/dev/null
python3 tools/pmc_traffic.py $OUT/pmc B32_N2048_L14_f16x3 $OUT/traffic.json "profiles/${TAG}_pmc_pass4.txt + ${TAG}_pmc_pass5.txt" > /dev/null 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --layers 63 --no-extra --no-cpu-baseline > $OUT/bench_L63.json 2>> $OUT/bench_default.err
python3 bench.py --layers 15 --no-extra --no-cpu-baseline > $OUT/bench_L15.json 2>> $OUT/bench_default.err
python3 bench.py --config cfg3 --no-extra > $OUT/bench_cfg3.json 2>> $OUT/bench_default.err
python3 bench.py --config cfg5 > $OUT/bench_cfg5.json 2>> $OUT/bench_default.err
python3 bench.py --no-graph --no-extra --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench_default.err
bash tools/train_prof.sh ${TAG}_prof/train > /dev/null 2>&1
bash tools/pmc_train.sh ${TAG}_prof/pmc_train > /dev/null 2>&1
ls -la $OUT $OUT/train
# r04: the rank-sized batch (4 clouds: what a rank of an 8-GPU job holds of BASELINE's 32) -- kernel trace and the two SQ counter
# passes of the 16-point-tile stack (csrc/flow16.hip) and the small-batch Chamfer scan
bash tools/kt_run.sh ${TAG}_prof/kernel_trace_stats_b4 --batch 4 --no-cpu-baseline --no-extra --pipelined 0 --steps 300 --warmup 50 --settle 100 > /dev/null 2>&1
bash tools/pmc_run.sh ${TAG}_prof/pmc_b4 --batch 4 > /dev/null 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_b4 B4_N2048_L14_f16x3 $OUT/traffic.json "profiles/${TAG}_pmc_b4_pass4.txt + ${TAG}_pmc_b4_pass5.txt" > /dev/null 2>&1
# r05: approx-EMD at cfg5 -- kernel trace and the two HBM counter passes of the matrix-core path
bash tools/kt_run.sh ${TAG}_prof/emd_kernel_trace --config cfg5 --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/pmc_cfg5.sh ${TAG}_prof/pmc_cfg5 > /dev/null 2>&1
python3 tools/pmc_traffic_emd.py $OUT/pmc_cfg5 16 8192 $OUT/traffic.json "profiles/${TAG}_pmc_cfg5_pass1.txt + _pass2.txt" > /dev/null 2>&1
# r05: the rank-sized training step (B = 8, G = 512) and the phase stamps of the training kernels
python3 bench.py --leg train --batch 8 --latent 512 --steps 40 --warmup 12 > $OUT/bench_train_B8_G512.json 2>> $OUT/bench_default.err
(cd dpf_nets_amd/csrc && make -s prof > /dev/null 2>&1)
python3 tools/train_kprof.py > $OUT/train_kprof.txt 2>&1
python3 tools/train_phase_prof.py > $OUT/train_phase_prof.txt 2>&1
ls -la $OUT
