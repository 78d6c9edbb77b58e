#!/bin/bash
# Run ON THE GPU BOX: the bench's KD step + per-GPU-batch proxies under several values of one environment switch:
#   bash scripts/sweep_env_step.sh VAR v1 v2 v3 ...
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
FLAGS="--steps 30 --warmup 8 --no-cpu-baseline --no-full-iteration --sweep 0 --no-config3 --no-roofline"
for v in "$@" "$1"; do
  env $VAR=$v python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'], {k: x.get('graph_ms') for k, x in d.get('strong_scaling_proxy_1gpu', {}).items()})"
done
