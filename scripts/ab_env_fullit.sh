#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of the bench's single-stream legs (full training iteration, saliency sweep) under an environment switch
VAR=$1; A=$2; B=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
FLAGS="--steps 10 --warmup 3 --no-cpu-baseline --sweep 3 --no-config3 --no-roofline --no-proxy"
for round in 1 2; do for v in $A $B; do
  env $VAR=$v python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['full_iteration']; print('$VAR=$v full_iteration', f['value'], f['ms_per_iteration'], 'sweep', d['saliency_sweep']['value'], 'step', d['ms_per_step'])"
done; done
