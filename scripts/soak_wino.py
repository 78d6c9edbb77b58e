"""Race soak: the Winograd / stride-2 / weight-gradient kernels have no atomics on these shapes, so every repetition must be
bit-identical to the first.  python scripts/soak_wino.py   (CAGC_WINO_NH=1|2 forces the workgroup shape)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
torch.manual_seed(0)
N = int(os.environ.get("REPS", "200"))
bad = 0
for (B, Cin, Cout, H) in [(16, 512, 512, 64), (16, 128, 128, 256), (4, 154, 154, 64), (2, 77, 39, 128), (16, 256, 256, 128)]:
    x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    s = torch.rand(B, Cin, device="cuda") + 0.5
    up = mc.pack_wino(w, 1.0, False)
    out = torch.empty(B, Cout, H, H, device="cuda"); first = None; diff = 0
    for i in range(N):
        out.fill_(float("nan"))
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), _lib.ptr(s), B, Cin, Cout, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
        if first is None: first = out.clone()
        elif not torch.equal(out, first): diff += 1
    print(f"wino {B}x{Cin}->{Cout}@{H}: {N} reps, {diff} differ, finite {bool(torch.isfinite(first).all())}", flush=True)
    bad += diff + (0 if torch.isfinite(first).all() else 1)
    # weight gradient (slab reduce: deterministic)
    g = torch.randn(B, Cout, H, H, device="cuda")
    gw = torch.empty(Cout, Cin, 3, 3, device="cuda")
    ws = torch.empty(_lib.query("cagc_modconv_wgrad_workspace", B, Cin, Cout, H, H, 3, 0), device="cuda")
    first = None; diff = 0
    for i in range(max(N // 10, 5)):
        _lib.call("cagc_modconv_wgrad", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(g), _lib.ptr(x), _lib.ptr(s), B, Cin, Cout, H, H, 3, 0, 1.0)
        if first is None: first = gw.clone()
        elif not torch.equal(gw, first): diff += 1
    print(f"wgrad {B}x{Cin}->{Cout}@{H}: {diff} differ, finite {bool(torch.isfinite(first).all())}", flush=True)
    bad += diff + (0 if torch.isfinite(first).all() else 1)
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
