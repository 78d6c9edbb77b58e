"""Timing of the 4x4 FIR passes of the discriminator (cagc_fir4x4_pitched) and of the blur behind the transposed conv (cagc_blur_up_fwd) at
the bench shapes; A/B for the former: CAGC_FIR_ROWS=0 (tiled LDS kernel).  A row-streaming cagc_blur_up_fwd was measured with the second
half of this script and not kept (0.776 vs 0.741 ms over these shapes)."""
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
print("cagc_blur_up_fwd (styled epilogue: d, per-image noise, bias, lrelu)")
tot = 0.0
fir = k.flatten().contiguous()
for (B, C, H) in [(16, 128, 128), (16, 256, 64), (16, 39, 128), (16, 512, 32), (16, 77, 64), (16, 154, 32), (2, 128, 128), (2, 512, 32)]:
    W = H; P = (W + 1 + 3) & ~3
    t_ = torch.randn(B, C, 4, H + 1, P, device="cuda"); out = torch.empty(B, C, 2 * H, 2 * W, device="cuda")
    d = torch.rand(B, C, device="cuda"); nz = torch.randn(B, 1, 2 * H, 2 * W, device="cuda"); nw = torch.ones(1, device="cuda"); bias = torch.randn(C, device="cuda")
    tm = timeit(lambda: _lib.call("cagc_blur_up_fwd", _lib.ptr(out), _lib.ptr(t_), _lib.ptr(fir), _lib.ptr(d), _lib.ptr(nz), B, _lib.ptr(nw), _lib.ptr(bias), B, C, H, W, 0.2, 1.4142135))
    by = 4.0 * B * C * (4 * (H + 1) * (W + 1) + 4 * H * W)
    tot += tm
    print(f"B {B} C {C} H {H}: {tm*1e6:7.1f} us  {by/tm/1e12:5.2f} TB/s")
print(f"sum {tot*1e3:.3f} ms")
