"""The saliency-sweep leg of bench.py (configs[4]: full 256 px generator fwd + bwd at bs 64) alone in one process — the leg is bimodal
per process on some boxes.  Prints img/s and, per kernel symbol, the total GPU time (HIP events around every C-ABI call through
bench.KernelTimer) so that a slow and a fast process can be diffed.   python scripts/sweep_modes.py [batches]"""
import os, sys, time, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
import bench
from cagc import kd, prune, _lib
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
student, teacher, disc = kd.build_synthetic_workload(256, dev, seed=0)
del student, disc
for p in teacher.parameters(): p.requires_grad_(True)
teacher.train()
mfn = lambda im: kd.ellipse_mask(im.shape[0], 256, dev)
prune.content_aware_scores(teacher, 64, 64, 0.05, mfn, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
prune.content_aware_scores(teacher, 64 * nb, 64, 0.05, mfn, dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
import ctypes
segs = []
clk = torch.zeros(2, device=dev)
for seg in range(int(os.environ.get("SEGS", "0"))):      # consecutive timed segments of 3 batches with the in-kernel shader-clock probe
    clk.zero_(); _lib.load().cagc_set_clock_probe(ctypes.c_void_p(clk.data_ptr()))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    prune.content_aware_scores(teacher, 64 * 3, 64, 0.05, mfn, dev)
    torch.cuda.synchronize(); d1 = time.perf_counter() - t1
    _lib.load().cagc_set_clock_probe(None)
    segs.append((round(64 * 3 / d1, 1), round(float(clk[0] / clk[1].clamp(min=1)))))
if segs: print("segments (img/s, MHz):", segs)
with bench.KernelTimer(_lib) as kt:
    prune.content_aware_scores(teacher, 64 * 2, 64, 0.05, mfn, dev)
agg = kt.summary()
top = sorted(((v[1] / 2, k, v[0] // 2) for k, v in agg.items()), reverse=True)[:14]
ptrs = sorted(p.data_ptr() for p in teacher.parameters())
print(json.dumps({"img_s": round(64 * nb / dt, 1), "ms_per_batch": round(dt / nb * 1e3, 2), "first_param_ptr": hex(ptrs[0]),
                  "mem_alloc_MB": torch.cuda.memory_allocated() >> 20, "top_ms_per_batch": [(k, round(ms, 3), n) for ms, k, n in top]}))
