import os, sys, random, torch
sys.path[:0] = ["/root/repo", "/root/repo/content-aware-gan-compression_amd"]
from cagc import kd
dev = torch.device("cuda")
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
mask = kd.ellipse_mask(16, 256, dev)
step = kd.GraphedKDStep(student, teacher, disc, 16, mask, random_noise=True)
rng = random.Random(0)
for i in range(301):
    l = step.sample_and_step(16, mask, rng, None)
    if i % 50 == 0:
        torch.cuda.synchronize()
        pn = sum(p.detach().float().norm().item() ** 2 for p in student.parameters()) ** 0.5
        print(i, "g %.4f kd_l1 %.4f |params| %.3f finite %s" % (l["g"].item(), l["kd_l1_loss"].item(), pn, all(torch.isfinite(p).all().item() for p in student.parameters())), flush=True)
from cagc import _lib
torch.cuda.synchronize()
print("stream-K error word", _lib.get_tuning("up4_error"))
