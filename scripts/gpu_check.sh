#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash scripts/gpu_check.sh TAG'): the whole -m gpu suite, then the driver-style bench line, then the
# kernel traces (per-GPU batch 16 and 2) that profiles/ summarises.  Outputs under gpurun_out/.
TAG=${1:-check}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/${TAG}_pytest.log
timeout 1200 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 400 gpurun_out/bench_${TAG}.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_${TAG}.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], {k: v["graph_ms"] for k, v in d["strong_scaling_proxy_1gpu"].items()},
      d["deterministic_mode"], d["config3_1024"], d["full_iteration"]["value"], d["saliency_sweep"]["value"], d["cpu_baseline"])
PY
