"""bench.py's N > 1 branches, end to end, before an 8-GPU node sees them: the driver's own launch line
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) as two processes that share the one GPU of the
test box (CAGC_SINGLE_DEVICE=1, gloo instead of RCCL — same control flow: launch-mode calibration with its all-reduced
verdict, HIP-graph capture per rank with the flat gradient all-reduce between the graphs, eager DistributedDataParallel
with bucket hooks firing from the custom autograd nodes while the teacher runs on its side stream, barrier + MAX-reduced
timing, rank 0's JSON line).  Replaces the reference's nn.DataParallel (train.py:522-525)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, CAGC_SINGLE_DEVICE="1", CAGC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line from rank 0: " + r.stdout[-1500:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["calibrated", "eager", "graph"])
def test_bench_two_ranks_on_one_gpu(mode):
    extra = {"calibrated": [], "eager": ["--no-graph"], "graph": ["--graph"]}[mode]
    d = _run(extra, 29600 + (os.getpid() % 80) + {"calibrated": 0, "eager": 100, "graph": 200}[mode])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["per_gpu_batch"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "strong" and d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    if mode == "calibrated":
        assert set(d["config"]["launch_mode_calibration_ms"]) == {"graph", "eager"}
    else:
        assert d["config"]["launch_mode"] == mode
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1.0     # rank 0's per-kernel pass ran
    assert d["full_iteration"] is None and d["cpu_baseline"] is None                # N = 1 legs only
