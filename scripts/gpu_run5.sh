#!/bin/bash
TAG=${1:-r4_e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv1x1_gpu.py -x -q 2>&1 | tail -5
timeout 300 python scripts/time_1x1.py 2>&1 | tee gpurun_out/${TAG}_time_1x1.log | grep gemm1x1
BS=2 timeout 300 python scripts/time_1x1.py 2>&1 | tee gpurun_out/${TAG}_time_1x1_bs2.log | grep gemm1x1
timeout 600 python -m pytest "tests/test_train_iter.py::test_ddp_discriminator_two_ranks_equal_one_rank" -x -q 2>&1 | tail -30
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-iteration --no-config3 --sweep 0 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 300 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:v["graph_ms"] for k,v in d["strong_scaling_proxy_1gpu"].items()})
PY
