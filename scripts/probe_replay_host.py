"""Is a HIP-graph replay launch asynchronous on the host while the previous replay of the same executable graph is still running?
Times the host side of consecutive GraphedKDStep.replay() calls (no synchronisation in between) against the GPU step time."""
import os, sys, time, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import kd
dev = torch.device("cuda")
BS = int(os.environ.get("BS", "2"))
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
mask = kd.ellipse_mask(BS, 256, dev)
step = kd.GraphedKDStep(student, teacher, disc, BS, mask, random_noise=True)
rng = random.Random(0)
for _ in range(5): step.sample_and_step(BS, mask, rng, None)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter(); host = []
for _ in range(N):
    a = time.perf_counter(); step.sample_and_step(BS, mask, rng, None); host.append(time.perf_counter() - a)
t_submit = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"bs {BS}: GPU {t_all / N * 1e3:.3f} ms/step; host submission {t_submit / N * 1e3:.3f} ms/step "
      f"(per call min {min(host)*1e3:.3f} median {sorted(host)[N//2]*1e3:.3f} max {max(host)*1e3:.3f} ms)")
# the graph launch alone
g = getattr(step, "graph_fb", None)
if g is not None:
    torch.cuda.synchronize(); hs = []
    t0 = time.perf_counter()
    for _ in range(N):
        a = time.perf_counter(); g.replay(); hs.append(time.perf_counter() - a)
    ts = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"   graph.replay() only: GPU {ta / N * 1e3:.3f} ms/step; host {ts / N * 1e3:.3f} ms per launch (min {min(hs)*1e3:.3f} max {max(hs)*1e3:.3f})")
# how much of the step is the per-step host->device copy of the mixing index in front of the launch?
import types
torch.cuda.synchronize()
def timed(n, f):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
a = timed(60, lambda: step.sample_and_step(BS, mask, rng, None))
b = timed(60, lambda: step.graph_fb.replay())
c = timed(60, lambda: step.sample_and_step(BS, mask, rng, None))
d = timed(60, lambda: step.graph_fb.replay())
print(f"   sample_and_step {a:.3f} / {c:.3f} ms;  graph replay alone {b:.3f} / {d:.3f} ms")
# is there a gap between two replays?  capture TWO steps in one graph and compare the per-step time
try:
    g2 = torch.cuda.CUDAGraph()
    step._in_graph_comm = False
    with torch.cuda.graph(g2, pool=step.graph_fb.pool()):
        step._fwd_bwd(); step._flat_optim.step()
        step._fwd_bwd(); step._flat_optim.step()
    e = timed(30, lambda: g2.replay()) / 2
    f = timed(60, lambda: step.graph_fb.replay())
    print(f"   two steps in one graph: {e:.3f} ms per step;  one step per graph {f:.3f} ms")
except Exception as ex:
    print("   two-step capture failed:", repr(ex)[:200])
