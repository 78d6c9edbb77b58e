"""Same-process A/B of the fused-phase persistent kernel (csrc/conv_up4.hip, cagc_set_tuning("up4", 1)) against the per-parity
register-direct launches (conv_rd.hip, "up4" 0) on the layers of configs[1]: the teacher's transposed convs (cagc_modconv_up_fwd) and the
discriminator's stride-2 data gradients (cagc_conv3x3s2_dgrad), batch 16.   python scripts/time_up4.py"""
import os, sys, time, ctypes, torch
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
MODES = [("rd", dict(up4=0, up25=0)), ("up4", dict(up4=1, up25=0)), ("up25", dict(up4=1, up25=1))]
if os.environ.get("FORCE"):      # no launch threshold: every layer goes to the named kernel
    MODES = [(m, dict(kn, up4_min_ksteps=0, up25_min_ksteps=0)) for m, kn in MODES]
if os.environ.get("ONLY"):
    MODES = [m for m in MODES if m[0] in os.environ["ONLY"].split(",")]
tot = {m: 0.0 for m, _ in MODES}
for (cin, cout, H) in [(512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128)]:
    wt = torch.randn(1, cout, cin, 3, 3, device="cuda")
    wp_fwd, _, _ = mc.pack_weights(wt, True)
    x, s = torch.randn(B, cin, H, H, device="cuda"), torch.rand(B, cin, device="cuda") + 0.5
    P = _lib.query("cagc_phase_pitch", H)
    t = torch.empty(B, cout, 4, H + 1, P, device="cuda")
    fl = 2.0 * B * cin * cout * 9 * H * H
    row = f"up_fwd {cin}->{cout} @{H}^2:"
    for name, kn in MODES:
        with _lib.tuning(**kn):
            dt, c = timeit(lambda: _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H))
        tot[name] += dt
        row += f"   {name} {dt*1e6:8.1f} us {fl/dt/1e12:6.1f} TF @{c:.0f} MHz"
    print(row, flush=True)
for (cin, cout, H) in [(128, 256, 256), (256, 512, 128), (512, 512, 64), (512, 512, 32)]:
    hb = H + 1; pitch = (hb + 3) // 4 * 4; ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    _, wp_bwd = mc.pack_plain_weights(w, 0.01, True)
    g = torch.randn(B, cout, ho, ho, device="cuda"); gx = torch.empty(B, cin, hb, pitch, device="cuda")
    fl = 2.0 * B * cin * cout * 9 * ho * ho
    row = f"s2 dgrad {cin}<-{cout} @{H}^2:"
    for name, kn in MODES:
        with _lib.tuning(**kn):
            dt, c = timeit(lambda: _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gx), _lib.ptr(g), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch))
        tot[name] += dt
        row += f"   {name} {dt*1e6:8.1f} us {fl/dt/1e12:6.1f} TF @{c:.0f} MHz"
    print(row, flush=True)
print("sum: " + "   ".join(f"{m} {v*1e3:.3f} ms" for m, v in tot.items()) + f"   error word {_lib.get_tuning('up4_error')}   up25 launches {_lib.get_tuning('up25_launches')}")
