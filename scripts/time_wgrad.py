"""Per-layer timing of cagc_modconv_wgrad on the student's shapes (bs 16).  python scripts/time_wgrad.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
B = 16
# (Cin, Cout, H_in, up)
B = int(os.environ.get("BS", "16"))
layers = [(154, 154, 4, 0), (154, 154, 4, 1), (154, 154, 8, 0), (154, 154, 8, 1), (154, 154, 16, 0), (154, 154, 16, 1),
          (154, 154, 32, 0), (154, 154, 32, 1), (154, 154, 64, 0), (154, 77, 64, 1), (77, 77, 128, 0), (77, 39, 128, 1),
          (39, 39, 256, 0)]
tot = 0.0
tot0 = 0.0
if os.environ.get("LAYERS"):      # e.g. LAYERS=8,10,12 : only these rows (PMC runs average per kernel symbol)
    layers = [layers[int(i)] for i in os.environ["LAYERS"].split(",")]
for cin, cout, H, up in layers:
    W = H
    x = torch.randn(B, cin, H, W, device="cuda")
    s = torch.rand(B, cin, device="cuda") + 0.5
    if up:
        P = _lib.query("cagc_phase_pitch", W)
        g = torch.randn(B, cout, 4, H + 1, P, device="cuda")
    else:
        g = torch.randn(B, cout, H, W, device="cuda")
    gw = torch.empty(cout, cin, 3, 3, device="cuda")
    ws = torch.empty(_lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, 3, up), device="cuda")
    def run():
        _lib.call("cagc_modconv_wgrad", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(g), _lib.ptr(x), _lib.ptr(s), B, cin, cout, H, W, 3, up, 1.0)
    res = {}
    for mode in ("pf", "nopf"):
        if mode == "nopf":   # baseline: generic row-indexed staging map, no register prefetch
            os.environ["CAGC_WGRAD_NOPF"] = "1"
            os.environ["CAGC_WGRAD_NOAFF"] = "1"
        else:
            os.environ.pop("CAGC_WGRAD_NOPF", None)
            os.environ.pop("CAGC_WGRAD_NOAFF", None)
        for _ in range(3): run()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t) / 20, gw.clone())
    os.environ.pop("CAGC_WGRAD_NOPF", None)
    os.environ.pop("CAGC_WGRAD_NOAFF", None)
    dt = res["pf"][0]
    same = torch.equal(res["pf"][1], res["nopf"][1])
    fl = 2.0 * B * cin * cout * 9 * H * W
    tot += dt
    tot0 += res["nopf"][0]
    print(f"cin {cin:3d} cout {cout:3d} H {H:3d} up {up}: {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF  (round-1 staging {res['nopf'][0]*1e6:8.1f} us) identical={same}  ws {ws.numel()*4/1e6:.1f} MB")
print(f"total {tot*1e3:.3f} ms   (round-1 staging {tot0*1e3:.3f} ms)")
