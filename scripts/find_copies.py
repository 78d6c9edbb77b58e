"""Which Python frames issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) in an eager KD step?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import kd
import random
student, teacher, disc = kd.build_synthetic_workload(256, "cuda", seed=0)
step = kd.KDStep(student, teacher, disc)
mask = kd.ellipse_mask(16, 256, "cuda")
rng = random.Random(0)
for _ in range(3): step.sample_and_step(16, mask, rng, None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step.sample_and_step(16, mask, rng, None)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous") and ev.device_time_total > 20:
        print(ev.name, round(ev.device_time_total), ev.input_shapes, [s for s in (ev.stack or []) if "repo" in s][:3])
