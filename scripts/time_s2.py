"""Per-layer timing of the discriminator's stride-2 3x3 convs (fwd + dgrad) at bs 16.  python scripts/time_s2.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if os.environ.get("LIB"):      # A/B against another build of the library (e.g. an ablation build)
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
from cagc.op import modconv as mc
B = int(os.environ.get("BS", "16"))
import ctypes
ZERO = os.environ.get("ZERO") == "1"     # same instruction stream on zero operands: what the board's power limit costs
clk = torch.zeros(2, device="cuda")
def timeit(f):
    """(seconds per launch, shader clock in MHz that workgroup 0 of these back-to-back launches saw: cagc_set_clock_probe)"""
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    clk.zero_(); _lib.load().cagc_set_clock_probe(ctypes.c_void_p(clk.data_ptr()))
    for _ in range(5): f()
    torch.cuda.synchronize(); _lib.load().cagc_set_clock_probe(None)
    return dt, float(clk[0] / clk[1].clamp(min=1))
for (cin, cout, H) in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32), (512, 512, 16), (512, 512, 8)]:
    hb = H + 1; pitch = (hb + 3) // 4 * 4; ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    if ZERO: w.zero_()
    wp_fwd, wp_bwd = mc.pack_plain_weights(w, 0.01, True)
    tmp = torch.randn(B, cin, hb, pitch, device="cuda"); out = torch.empty(B, cout, ho, ho, device="cuda")
    g = torch.randn(B, cout, ho, ho, device="cuda"); gtmp = torch.empty(B, cin, hb, pitch, device="cuda")
    if ZERO: tmp.zero_(); g.zero_()
    (tf, cf) = timeit(lambda: _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(tmp), _lib.ptr(wp_fwd), B, cin, cout, hb, hb, pitch))
    (tb, cb) = timeit(lambda: _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gtmp), _lib.ptr(g), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch))
    fl = 2.0 * B * cin * cout * 9 * ho * ho
    print(f"cin {cin} cout {cout} H {H}: fwd {tf*1e6:8.1f} us {fl/tf/1e12:6.1f} TF @{cf:.0f} MHz   dgrad {tb*1e6:8.1f} us {fl/tb/1e12:6.1f} TF @{cb:.0f} MHz")
