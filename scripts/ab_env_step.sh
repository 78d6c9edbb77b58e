#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of the bench's KD step under an environment switch:  bash scripts/ab_env_step.sh VAR A B
VAR=$1; A=$2; B=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
FLAGS="--steps 40 --warmup 10 --no-cpu-baseline --no-full-iteration --sweep 0 --no-config3 --no-roofline"
for round in 1 2; do for v in $A $B; do
  env $VAR=$v python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'], {k: x.get('graph_ms') for k, x in d.get('strong_scaling_proxy_1gpu', {}).items()})"
done; done
