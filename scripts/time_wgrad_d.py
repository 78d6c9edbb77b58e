"""Weight gradients of the discriminator's 3x3 layers (D training step) at bs 16: stride-1 (cagc_modconv_wgrad, s = null)
and stride-2 (phase-planar role swap).  CAGC_WGRAD_NO4=1 disables the 64-channel plans.  python scripts/time_wgrad_d.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc.op import conv_closure as cc
B = 16
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
tot = 0.0
for cin, cout, H in [(128, 128, 256), (256, 256, 128), (512, 512, 64), (512, 512, 32), (512, 512, 16)]:
    x = torch.randn(B, cin, H, H, device="cuda"); g = torch.randn(B, cout, H, H, device="cuda")
    t1 = bench(lambda: cc.wgrad_s1(g, x, 3, 0.1))
    fl = 2.0 * B * cin * cout * 9 * H * H
    print(f"s1 {cin:3d}->{cout:3d} @{H:3d}^2: {t1*1e6:8.1f} us  {fl/t1/1e12:6.1f} TF")
    tot += t1
for cin, cout, H in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32)]:
    ho = H // 2
    xb = torch.randn(B, cin, 2 * ho + 1, 2 * ho + 1, device="cuda"); g = torch.randn(B, cout, ho, ho, device="cuda")
    t2 = bench(lambda: cc.wgrad_s2(g, xb, 0.1))
    fl = 2.0 * B * cin * cout * 9 * ho * ho
    print(f"s2 {cin:3d}->{cout:3d} @{H:3d}^2: {t2*1e6:8.1f} us  {fl/t2/1e12:6.1f} TF")
    tot += t2
print(f"total {tot*1e3:.2f} ms")
