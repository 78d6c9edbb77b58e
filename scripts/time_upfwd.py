import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if os.environ.get("LIB"):      # A/B against another build of the library (e.g. an ablation build)
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
from cagc.op import modconv as mc
B = int(os.environ.get("BS", "16"))
_only = os.environ.get("ONLY")      # e.g. ONLY=512,256,64 : a single layer (PMC runs)
for (cin, cout, H) in ([tuple(int(v) for v in _only.split(","))] if _only else [(512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128), (154, 154, 4), (154, 154, 8), (154, 154, 16), (154, 154, 32), (154, 77, 64), (77, 39, 128)] if len(sys.argv) < 2 else [(512, 512, 4), (512, 512, 8), (512, 512, 16), (154, 154, 8), (154, 154, 16)]):
    x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(1, cout, cin, 3, 3, device="cuda")
    s = torch.rand(B, cin, device="cuda") + 0.5
    wp_fwd, wp_bwd, wsq = mc.pack_weights(w, False)
    P = _lib.query("cagc_phase_pitch", H)
    t = torch.empty(B, cout, 4, H + 1, P, device="cuda")
    def run():
        _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ph = os.environ.get("CAGC_UP_PHASE")
    taps = {None: 9, "0": 4, "1": 2, "2": 2, "3": 1}[ph]
    fl = 2.0 * B * cin * cout * taps * H * H
    print(f"phase {ph} cin {cin} cout {cout} H {H}: {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF")
