#!/bin/bash
# ms per step of the training leg at a rank-sized batch (B = 8, G = 512) for several builds / environments on the SAME box:
#   tools/train_ab_small.sh <so> [<so> ...]      (each once with DPF_TRAIN_ROLES unset, =0 and =1)
for so in "$@"; do
  for r in "" 0 1; do
    if [ -n "$r" ]; then export DPF_TRAIN_ROLES=$r; else unset DPF_TRAIN_ROLES; fi
    python3 $GRAFT_REPO_ROOT/tools/lib_ab.py $so --leg train --batch 8 --latent 512 --steps 40 --warmup 12 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print('$so', 'ROLES=$r', 'ms/step %.3f' % d['ms_per_step'], {n: round(v['us_per_layer'], 2) for n, v in k.items() if isinstance(v, dict) and v['us_per_layer'] > 1})"
  done
done
