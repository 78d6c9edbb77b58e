"""Timing of the 4x4 FIR passes of the discriminator (cagc_fir4x4_pitched) at the bench shapes; A/B: CAGC_FIR_ROWS=0 (tiled LDS kernel)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 20
k1 = torch.tensor([1., 3., 3., 1.]); k = (k1[:, None] * k1[None, :]); k = (k / k.sum()).cuda()
tot = 0.0
for (planes, ih, iw, ip, oh, ow, op, pad) in [(2048, 256, 256, 256, 257, 257, 260, 2), (2048, 257, 257, 260, 256, 256, 256, 1),
                                             (4096, 128, 128, 128, 129, 129, 132, 2), (4096, 129, 129, 132, 128, 128, 128, 1),
                                             (8192, 64, 64, 64, 65, 65, 68, 2), (8192, 65, 65, 68, 64, 64, 64, 1),
                                             (8192, 32, 32, 32, 33, 33, 36, 2)]:
    x = torch.randn(planes, ih, ip, device="cuda"); out = torch.empty(planes, oh, op, device="cuda")
    t = timeit(lambda: _lib.call("cagc_fir4x4_pitched", _lib.ptr(out), _lib.ptr(x), _lib.ptr(k), planes, ih, iw, ip, oh, ow, op, pad, pad))
    by = 4.0 * planes * (ih * iw + oh * ow)
    tot += t
    print(f"planes {planes} {ih}x{iw}/{ip} -> {oh}x{ow}/{op}: {t*1e6:7.1f} us  {by/t/1e12:5.2f} TB/s")
print(f"sum {tot*1e3:.3f} ms")
