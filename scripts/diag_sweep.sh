#!/bin/bash
# Run ON THE GPU BOX: is the saliency-sweep leg of bench.py stable with every other leg on (driver-style default run minus the CPU baseline)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('step', d['ms_per_step'], 'sweep', d['saliency_sweep']['value'], 'full_it', d['full_iteration']['value'], 'c3', d['config3_1024']['value'], 'clock', r['shader_clock_mhz'], 'graph clock', (r.get('graph_replay_clock') or {}).get('mhz'))"
done
