#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash scripts/final_measure.sh TAG'): the driver-style default bench line, then the kernel trace + PMC passes
# (scripts/profile_step.sh) at per-GPU batch 16 and the kernel trace at per-GPU batch 2.  No pytest here (scripts/gpu_check.sh runs the suite).
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}.json"))
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, "roofline", r["kernel"][:40], r["achieved"], r["frac"], "clock", r.get("shader_clock_mhz"), r.get("frac_at_measured_clock"), "avg_launch_ms", r["avg_launch_ms"])
print({k: v["graph_ms"] for k, v in d["strong_scaling_proxy_1gpu"].items()}, d["deterministic_mode"], d["config3_1024"], d["full_iteration"]["value"], d["saliency_sweep"]["value"], d["cpu_baseline"])
PY
bash scripts/profile_step.sh ${TAG} 2>&1 | tail -16
PROFILE_EXTRA="--local-batch 2" PROFILE_KT_ONLY=1 bash scripts/profile_step.sh ${TAG}_bs2 2>&1 | tail -8
