#!/bin/bash
# Run ON THE GPU BOX: the saliency-sweep leg with and without the clock probe, interleaved, own process each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for rep in 1 2 3 4 5; do for f in "" "--no-sweep-clock"; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-iteration --no-config3 --no-proxy --no-roofline --sweep 10 $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['saliency_sweep']
print('probe', 'off' if '$f' else 'on ', ' sweep', s['value'], 'img/s  @', s['shader_clock_mhz'], 'MHz')"
done; done
