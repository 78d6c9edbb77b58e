#!/bin/bash
TAG=${1:-r4_d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv1x1_gpu.py tests/test_banked_ops_gpu.py -x -q 2>&1 | tail -15
timeout 300 python scripts/time_1x1.py 2>&1 | tee gpurun_out/${TAG}_time_1x1.log | tail -14
BS=2 timeout 300 python scripts/time_1x1.py 2>&1 | tee gpurun_out/${TAG}_time_1x1_bs2.log | grep gemm1x1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_train_iter.py tests/test_second_order_gpu.py tests/test_kd_gates_gpu.py tests/test_bench_selection_gpu.py -x -q 2>&1 | tail -6
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-iteration --no-config3 --sweep 0 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 300 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:v["graph_ms"] for k,v in d["strong_scaling_proxy_1gpu"].items()})
for k,v in list(d["roofline"]["cagc_kernel_ms_per_step"].items())[:40]: print(k, v)
PY
