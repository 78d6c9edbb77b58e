"""Per-layer timing of the student's weight gradients (cagc_modconv_wgrad_demod: register-direct kernel + slab reduce) at the
student's pruned widths.  BS=16 python scripts/time_wgrad.py"""
import os, sys, time, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if os.environ.get("LIB"):      # A/B against another build of the library
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
B = int(os.environ.get("BS", "16"))
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 10
tot = 0.0
ONLY = os.environ.get("ONLY")
# (cin, cout, H of the conv's INPUT, up)
for (cin, cout, H, up) in [(154, 154, 16, 0), (154, 154, 16, 1), (154, 154, 32, 0), (154, 154, 32, 1), (154, 154, 64, 0), (154, 77, 64, 1), (77, 77, 128, 0), (77, 39, 128, 1), (39, 39, 256, 0)]:
    if ONLY and ONLY != f"{cin},{cout},{H},{up}":
        continue
    W = H
    x = torch.randn(B, cin, H, W, device="cuda"); s = torch.randn(B, cin, device="cuda")
    if up:
        pitch = (W + 1 + 3) & ~3
        g = torch.randn(B, cout, 4, H + 1, pitch, device="cuda")
    else:
        g = torch.randn(B, cout, H, W, device="cuda")
    n_ws = _lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, 3, up)
    ws = torch.empty(n_ws, device="cuda"); gw = torch.empty(1, cout, cin, 3, 3, device="cuda")
    t = timeit(lambda: _lib.call("cagc_modconv_wgrad_demod", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(g), _lib.ptr(x), _lib.ptr(s), None, None,
                                 B, cin, cout, H, W, 3, up, 1.0 / math.sqrt(cin * 9)))
    fl = 2.0 * B * cin * cout * 9 * H * W
    tot += t
    print(f"cin {cin:4d} cout {cout:4d} H {H:4d} up {up}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF  (ws {n_ws*4/1e6:.1f} MB)")
print(f"sum {tot*1e3:.3f} ms")
