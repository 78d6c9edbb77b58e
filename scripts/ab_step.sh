#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of two builds on the bench's KD step (cagc/libcagc_hip.so vs cagc/libcagc_hip_old.so), short legs only.
TAG=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=content-aware-gan-compression_amd/cagc
FLAGS="--steps 40 --warmup 10 --no-cpu-baseline --no-full-iteration --sweep 0 --no-config3 --no-roofline"
cp $L/libcagc_hip.so /tmp/new.so
for round in 1 2; do
  for which in new old; do
    [ $which = old ] && cp $L/libcagc_hip_old.so $L/libcagc_hip.so || cp /tmp/new.so $L/libcagc_hip.so
    python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', d['value'], d['ms_per_step'], {k: v.get('graph_ms') for k, v in d.get('strong_scaling_proxy_1gpu', {}).items()})"
  done
done | tee gpurun_out/${TAG}_step.log
cp /tmp/new.so $L/libcagc_hip.so
