import os
os.environ.setdefault("CAGC_WINO4_MIN_WGS", "0")   # parity on small launches: keep them on the F(4x4) kernel
"""F(4x4,3x3) Winograd kernel vs float64: plain linear / styled epilogue / gated data gradient, and timing against F(2x2).
CAGC_WINO_F4=0|1 python scripts/check_wino4.py"""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
dev = "cuda"
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
torch.manual_seed(0)
for (B, cin, cout, H, W) in [(1, 128, 128, 8, 32), (2, 136, 128, 16, 32), (2, 256, 256, 32, 64), (1, 512, 512, 64, 64)]:
    x = torch.randn(B, cin, H, W); w = torch.randn(cout, cin, 3, 3); s = torch.rand(B, cin) + 0.5; d = torch.rand(B, cout) + 0.5
    noise = torch.randn(B, 1, H, W); nw = torch.tensor([0.3]); bias = 0.1 * torch.randn(cout)
    scale = 1.0 / (cin * 9) ** 0.5
    xg, sg, dg, ng, nwg, bg = (t.to(dev) for t in (x, s, d, noise, nw, bias))
    up = mc.pack_wino(w.to(dev), scale, False)
    lin = F.conv2d(x.double() * s.double()[:, :, None, None], w.double() * scale, padding=1) * d.double()[:, :, None, None]
    out = torch.full((B, cout, H, W), float("nan"), device=dev)
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 0, _lib.ptr(dg), None, 0, None, None, 0.2, 1.0)
    e_lin = rel(out, lin)
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 1, _lib.ptr(dg), _lib.ptr(ng), B, _lib.ptr(nwg), _lib.ptr(bg), 0.2, 2 ** 0.5)
    ref = F.leaky_relu(lin + 0.3 * noise.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    e_sty = rel(out, ref)
    # gated data gradient: gx = conv_transpose(gout * lrelu'(act)) + residual
    gout = torch.randn(B, cout, H, W); act = torch.randn(B, cout, H, W); res = torch.randn(B, cin, H, W)
    upb = mc.pack_wino(w.to(dev), scale, True)
    gx = torch.full((B, cin, H, W), float("nan"), device=dev)
    gg, ag, rg = gout.to(dev), act.to(dev), res.to(dev)
    _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(ag), _lib.ptr(upb), _lib.ptr(rg), B, cin, cout, H, W, 0.2, 2 ** 0.5)
    gin = gout.double() * torch.where(act > 0, 1.0, 0.2).double() * 2 ** 0.5
    refg = F.conv_transpose2d(gin, w.double() * scale, padding=1) + res.double()
    e_g = rel(gx, refg)
    print(f"B{B} {cin}->{cout} {H}x{W}: linear {e_lin:.2e}  styled {e_sty:.2e}  gated dgrad {e_g:.2e}")
for (B, C, H) in [(16, 512, 64), (16, 256, 128), (16, 128, 256), (16, 512, 32)]:
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(C, C, 3, 3, device=dev)
    up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
    def run():
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    fl = 2.0 * B * C * C * 9 * H * H
    print(f"time B{B} C{C} H{H}: {dt*1e3:.3f} ms  direct-equiv {fl/dt/1e12:.1f} TF")
