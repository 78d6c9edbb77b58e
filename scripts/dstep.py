"""A few discriminator steps (TrainIteration.d_step, bs 16) for rocprofv3:  rocprofv3 --kernel-trace --stats -- python scripts/dstep.py"""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import kd
dev = torch.device("cuda")
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
it = kd.TrainIteration(student, teacher, disc)
B = 16
real = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
what = sys.argv[1] if len(sys.argv) > 1 else "d"
N = int(os.environ.get("N", "4"))
for i in range(2 + N):
    if what == "d":
        it.d_step(real, [torch.randn(B, 512, device=dev)])
    elif what == "r1":
        it.d_reg(real)
    else:
        it.g_reg([torch.randn(B // 2, 512, device=dev)])
torch.cuda.synchronize()
print("done", what)
