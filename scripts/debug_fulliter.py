import os, sys, time, copy, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import kd, _lib
import cagc.op.upfirdn2d as U
orig = _lib.call
def traced(name, *a):
    if name == "cagc_upfirdn2d":
        print("   upfirdn2d planes", a[3], "in", a[4], a[5], "out", a[6], a[7], "k", a[8], a[9], "up", a[10], a[11], "down", a[12], a[13], "pad", a[14:18], flush=True)
    else:
        print("  ", name, flush=True)
    r = orig(name, *a)
    torch.cuda.synchronize(); print("   ok", flush=True)
    return r
_lib.call = traced
dev = torch.device("cuda")
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
g_ema = copy.deepcopy(student)
it = kd.TrainIteration(student, teacher, disc, g_ema=g_ema)
bs = 16
mask = kd.ellipse_mask(bs, 256, dev)
real = torch.rand(bs, 3, 256, 256, device=dev) * 2 - 1
rng = random.Random(0)
for i in (1, 0):
    print("iteration", i, flush=True)
    it.iteration(i, real, mask, rng, None)
    torch.cuda.synchronize()
    print("done", i, flush=True)
