"""Per-(entry point, shape) GPU time of one eager KD step at the bench workload (HIP events around every libcagc call).
python scripts/per_layer.py [min_us]"""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
import bench
from cagc import _lib, kd
if os.environ.get("LIB"):      # a timing-only build variant (scripts/build_variant.sh)
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
dev = torch.device("cuda")
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
BS = int(os.environ.get("BS", "16"))
mask = kd.ellipse_mask(BS, 256, dev)
step = kd.KDStep(student, teacher, disc)
rng = random.Random(0)
for _ in range(2): step.sample_and_step(BS, mask, rng, None)
orig = _lib.call
recs = []
def timed(name, *args):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); orig(name, *args); e.record()
    ints = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and a < 100000)
    recs.append((name, ints[:7], s, e, bench.conv_flops(name, args)))
_lib.call = timed
N = 3
for _ in range(N): step.sample_and_step(BS, mask, rng, None)
torch.cuda.synchronize()
agg = {}
for name, key, s, e, fl in recs:
    d = agg.setdefault((name, key), [0, 0.0, 0.0]); d[0] += 1; d[1] += s.elapsed_time(e); d[2] += fl
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
tot = sum(v[1] for v in agg.values()) / N
print(f"total libcagc event time {tot:.2f} ms/step")
for (name, key), (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = ms / cnt * 1e3
    if ms / N * 1e3 < thr: continue
    tf = f"{fl / (ms * 1e-3) / 1e12:6.1f} TF" if fl > 0 else "        "
    print(f"{name:26s} {str(key):44s} x{cnt // N:2d}  {us:8.1f} us each  {ms / N:7.3f} ms/step  {tf}")
if os.environ.get("DUMP"):      # machine-readable: {"entry point (shape)": [launches per step, ms per step, flops per step]}
    import json
    json.dump({f"{name} {key}": [cnt / N, ms / N, fl / N] for (name, key), (cnt, ms, fl) in agg.items()}, open(os.environ["DUMP"], "w"))
