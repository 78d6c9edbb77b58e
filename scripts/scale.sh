#!/bin/bash
# The multi-GPU run of the headline bench, exactly as bench.py's docstring / the driver launch it, made boring:
#   bash scripts/scale.sh N [extra bench.py flags]        (N ranks on ONE node, one process per GPU, RCCL over xGMI)
# CAGC_STRICT_COMM=1: a collective that cannot be captured inside the step graph is an ERROR, not a silent change of mode.
# Asserts on rank 0's JSON line: RCCL saw N ranks, the gradient collective ran in the mode the line reports, and prints the step time next
# to the per-bucket all-reduce times (HIP events).  On a 1-GPU box: CAGC_SINGLE_DEVICE=1 CAGC_DIST_BACKEND=gloo bash scripts/scale.sh 2
set -euo pipefail
N=${1:?usage: scripts/scale.sh N [bench flags]}; shift || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 CAGC_STRICT_COMM=${CAGC_STRICT_COMM:-1}
OUT=${SCALE_OUT:-gpurun_out/scale_n${N}.json}
mkdir -p "$(dirname "$OUT")"
if [ "$N" -eq 1 ]; then
  python bench.py --gpus 1 --steps "${STEPS:-50}" --warmup "${WARMUP:-10}" "$@" | tail -1 > "$OUT"
else
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${PORT:-29511}" \
    bench.py --gpus "$N" --steps "${STEPS:-50}" --warmup "${WARMUP:-10}" "$@" | grep '^{' | tail -1 > "$OUT"
fi
N=$N BACKEND=${CAGC_DIST_BACKEND:-nccl} python - "$OUT" <<'PY'
import json, os, sys
d = json.load(open(sys.argv[1]))
n, c = int(os.environ["N"]), d["config"]
assert d["n_gpus"] == n and c["dist_world_size"] == n, (d["n_gpus"], c["dist_world_size"])
if n > 1:
    assert c["dist_initialized"] and c["dist_backend"] == os.environ["BACKEND"], c["dist_backend"]
    assert c["grad_collective"] in ("graph", "host", "ddp_buckets"), c["grad_collective"]
    if c["launch_mode"] == "graph" and os.environ["BACKEND"] == "nccl":
        assert c["grad_collective"] == "graph", f"collectives were not captured inside the step graph: {c.get('grad_collective_reason')}"
    assert c["grad_bucket_allreduce_ms"] and len(c["grad_bucket_allreduce_ms"]) == len(c["grad_bucket_bytes"])
print(f"N={n}  {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  launch_mode={c['launch_mode']}  backend={c['dist_backend']}  "
      f"grad_collective={c['grad_collective']} ({c.get('grad_collective_reason')})")
if n > 1:
    for i, (b, ms) in enumerate(zip(c["grad_bucket_bytes"], c["grad_bucket_allreduce_ms"])):
        print(f"  bucket {i}: {b / 1e6:7.2f} MB  all-reduce {ms * 1e3:8.1f} us  ({2 * (n - 1) / n * b / (ms * 1e-3) / 1e9:6.1f} GB/s bus)")
    print(f"  sum {c['grad_allreduce_ms_sum']:.3f} ms = {100 * c['grad_allreduce_ms_sum'] / d['ms_per_step']:.1f} % of the step if nothing hid it")
PY
