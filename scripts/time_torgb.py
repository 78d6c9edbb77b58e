"""Run ON THE GPU BOX: cagc_torgb_fwd on the ToRGB launches of configs[1] (teacher + student, batch 16):  python scripts/time_torgb.py [lib.so ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
libs = sys.argv[1:] or ["libcagc_hip.so"]
if len(libs) > 1:
    import subprocess
    for l in libs:
        print(l, subprocess.run([sys.executable, __file__, l], capture_output=True, text=True).stdout.strip().splitlines()[-1])
    sys.exit(0)
_lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", libs[0])
B = int(os.environ.get("B", 16))
tot, byts = 0.0, 0.0
for (C, H) in [(128, 256), (256, 128), (512, 64), (512, 32), (39, 256), (77, 128), (154, 64), (154, 32)]:
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(3, C, device="cuda"); s = torch.rand(B, C, device="cuda")
    bias = torch.zeros(3, device="cuda"); skip = torch.randn(B, 3, H // 2, H // 2, device="cuda"); fir = torch.ones(4, 4, device="cuda") / 4
    out = torch.empty(B, 3, H, H, device="cuda")
    f = lambda: _lib.call("cagc_torgb_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(w), _lib.ptr(s), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(fir), B, C, H, H, 0.1)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    by = 4.0 * B * H * H * (C + 3.75)
    tot += dt; byts += by
    print(f"C {C} H {H}: {dt*1e6:.1f} us {by/dt/1e12:.2f} TB/s")
print(f"sum {tot*1e3:.3f} ms  {byts/tot/1e12:.2f} TB/s")
