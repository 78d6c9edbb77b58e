"""Run ON THE GPU BOX: the ToRGB / skip forks in eager mode and under HIP-graph capture, with faulthandler."""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
import torch
from cagc import kd
B = int(os.environ.get("B", 2))
dev = torch.device("cuda:0")
student, teacher, disc = kd.build_synthetic_workload(256, dev)
mask = kd.ellipse_mask(B, 256, dev)
print("built", flush=True)
step = kd.KDStep(student, teacher, disc)
for i in range(3):
    out = step.sample_and_step(B, mask)
    torch.cuda.synchronize()
    print("eager step", i, float(out["g"]), float(out["kd_l1_loss"]), flush=True)
g = kd.GraphedKDStep(student, teacher, disc, B, mask)
print("captured", flush=True)
for i in range(3):
    out = g.sample_and_step()
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20):
    out = g.sample_and_step()
torch.cuda.synchronize()
print("graph ms", (time.perf_counter() - t) / 20 * 1e3, float(out["g"]), flush=True)
