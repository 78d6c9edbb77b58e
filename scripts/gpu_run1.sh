#!/bin/bash
# GPU run: new end-to-end parity tests, bench line, launch census and graph-replay timeline at per-GPU batch 2.
TAG=${1:-r4_a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bench_selection_gpu.py tests/test_rccl_world1_gpu.py "tests/test_gpu_parity.py::test_full_256_teacher_forward_vs_oracle" "tests/test_gpu_parity.py::test_graphed_kd_step_resumes_from_saved_optimizer_state_and_invalidates_frozen_caches" tests/test_wino4_gpu.py tests/test_kd_gates_gpu.py -x -q -s 2>&1 | tail -40 > gpurun_out/${TAG}_newtests.log
tail -15 gpurun_out/${TAG}_newtests.log
timeout 900 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 600 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["strong_scaling_proxy_1gpu"], d["deterministic_mode"], d["config3_1024"], d["full_iteration"]["value"], d["saliency_sweep"]["value"])
PY
timeout 600 python scripts/launch_census.py --local-batch 2 > gpurun_out/${TAG}_census_bs2.md 2>&1; head -50 gpurun_out/${TAG}_census_bs2.md
rm -rf /tmp/prof_g
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_g -o kt -- python bench.py --graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-full-iteration --no-proxy --no-config3 --sweep 0 --local-batch 2 > gpurun_out/${TAG}_graph_bs2.log 2>&1
DB=$(find /tmp/prof_g -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" --marker k_masked_l1 --last 4 --gaps 30 --dump > gpurun_out/${TAG}_graph_bs2_timeline.md 2>&1; head -45 gpurun_out/${TAG}_graph_bs2_timeline.md
