#!/bin/bash
# Run ON THE GPU BOX: KD step + per-GPU-batch proxies with the ToRGB side-stream chain on / off (CAGC_FORK_TORGB)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
FLAGS="--steps 30 --warmup 8 --no-cpu-baseline --no-full-iteration --sweep 0 --no-config3 --no-roofline"
for v in 0 2 0 2; do
  env CAGC_FORK_TORGB=$v python bench.py $FLAGS 2>gpurun_out/ab_fork_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TORGB=$v', d['value'], d['ms_per_step'], {k: (x.get('graph_ms'), x.get('eager_ms')) for k, x in d.get('strong_scaling_proxy_1gpu', {}).items()})"
done 2>&1 | tee gpurun_out/${1:-r06}_ab_fork.log
tail -5 gpurun_out/ab_fork_1.err
