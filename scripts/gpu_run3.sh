#!/bin/bash
TAG=${1:-r4_c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_banked_ops_gpu.py tests/test_bench_selection_gpu.py -q -s 2>&1 | tail -60 > gpurun_out/${TAG}_newtests.log
grep -v "^$" gpurun_out/${TAG}_newtests.log | grep -i "F(4x4)\|gates\|passed\|failed\|Error\|additivity\|assert" | cut -c1-600 | tail -20
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-full-iteration --no-config3 --sweep 0 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 300 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["strong_scaling_proxy_1gpu"], d["deterministic_mode"])
PY
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
