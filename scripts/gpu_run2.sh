#!/bin/bash
TAG=${1:-r4_b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/graph_probe.py > gpurun_out/${TAG}_graph_probe.md 2>&1; cat gpurun_out/${TAG}_graph_probe.md | tail -22
timeout 1500 python -m pytest tests/test_bench_selection_gpu.py tests/test_rccl_world1_gpu.py "tests/test_gpu_parity.py::test_full_256_teacher_forward_vs_oracle" "tests/test_gpu_parity.py::test_graphed_kd_step_resumes_from_saved_optimizer_state_and_invalidates_frozen_caches" -q -s 2>&1 | tail -40 > gpurun_out/${TAG}_newtests.log
grep -v "^$" gpurun_out/${TAG}_newtests.log | tail -25 | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-full-iteration --no-config3 --sweep 0 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 300 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["strong_scaling_proxy_1gpu"], d["deterministic_mode"])
PY
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
