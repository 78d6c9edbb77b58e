"""Same-process A/B of the Winograd-domain stride-2 forward kernel (csrc/conv_s2w.hip, cagc_set_tuning("s2w", 1)) against the direct
kernels (conv_rd.hip: k_conv_s2v / k_conv_rd) on the discriminator's down-sampling convs of configs[1].   python scripts/time_s2w.py"""
import os, sys, time, math, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
from cagc.op import modconv as mc
B = int(os.environ.get("BS", "16"))
REPS = int(os.environ.get("REPS", "10"))
clk = torch.zeros(2, device="cuda")
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(REPS): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / REPS
    clk.zero_(); _lib.load().cagc_set_clock_probe(ctypes.c_void_p(clk.data_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize(); _lib.load().cagc_set_clock_probe(None)
    return dt, float(clk[0] / clk[1].clamp(min=1))
MODES = [("direct", dict(s2w=0)), ("s2w", dict(s2w=1))]
if os.environ.get("FORCE"):
    MODES = [(m, dict(kn, s2w_min_ksteps=0)) for m, kn in MODES]
if os.environ.get("ONLY"):
    MODES = [m for m in MODES if m[0] in os.environ["ONLY"].split(",")]
tot = {m: 0.0 for m, _ in MODES}
for (cin, cout, H) in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32), (512, 512, 16)]:
    hb = H + 1; pitch = (hb + 3) // 4 * 4; ho = H // 2
    w = torch.randn(cout, cin, 3, 3, device="cuda"); bias = torch.randn(cout, device="cuda")
    wp_fwd, _ = mc.pack_plain_weights(w, 0.01, True)
    x = torch.randn(B, cin, hb, pitch, device="cuda"); out = torch.empty(B, cout, ho, ho, device="cuda")
    fl = 2.0 * B * cin * cout * 9 * ho * ho
    row = f"s2 fwd {cin}->{cout} @{H}^2:"
    for name, kn in MODES:
        with _lib.tuning(**kn):
            dt, c = timeit(lambda: _lib.call("cagc_conv3x3s2_act_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(bias), B, cin, cout, hb, hb, pitch, 0.2, math.sqrt(2.0)))
        tot[name] += dt
        row += f"   {name} {dt*1e6:8.1f} us {fl/dt/1e12:6.1f} TF @{c:.0f} MHz"
    print(row, flush=True)
# the student's up-layer data gradients (cagc_modconv_up_dgrad: the planar form of the same kernel, with the style-gradient reduction)
for (cin, cout, H) in [(77, 39, 128), (154, 77, 64), (154, 154, 32), (154, 154, 16)]:
    wt = torch.randn(1, cout, cin, 3, 3, device="cuda")
    _, wp_bwd, _ = mc.pack_weights(wt, True)
    P = _lib.query("cagc_phase_pitch", H)
    gt = torch.randn(B, cout, 4, H + 1, P, device="cuda"); x = torch.randn(B, cin, H, H, device="cuda"); sc = torch.rand(B, cin, device="cuda") + 0.5
    gx = torch.empty(B, cin, H, H, device="cuda"); gs = torch.zeros(B, cin, device="cuda")
    fl = 2.0 * B * cin * cout * 9 * H * H
    row = f"up dgrad {cin}<-{cout} @{H}^2:"
    for name, kn in MODES:
        with _lib.tuning(**kn):
            dt, c = timeit(lambda: _lib.call("cagc_modconv_up_dgrad", _lib.ptr(gx), _lib.ptr(gs), _lib.ptr(gt), _lib.ptr(wp_bwd), _lib.ptr(sc), _lib.ptr(x), B, cin, cout, H, H))
        tot[name] += dt
        row += f"   {name} {dt*1e6:8.1f} us {fl/dt/1e12:6.1f} TF @{c:.0f} MHz"
    print(row, flush=True)
print("sum: " + "   ".join(f"{m} {v*1e3:.3f} ms" for m, v in tot.items()) + f"   error word {_lib.get_tuning('up4_error')}   s2w launches {_lib.get_tuning('s2w_launches')}")
