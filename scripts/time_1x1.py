"""1x1 convs of the discriminator skip path: the implicit-GEMM kernel (cagc_modconv_fwd / _dgrad, k = 1) vs a plain rocBLAS
batched SGEMM (torch.matmul) on the same NCHW operands.  python scripts/time_1x1.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
B = int(os.environ.get("BS", "16"))
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for cin, cout, H in [(128, 256, 128), (256, 512, 64), (512, 512, 32), (512, 512, 16), (512, 512, 8), (512, 512, 4)]:
    x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda")
    g = torch.randn(B, cout, H, H, device="cuda")
    wp_fwd, wp_bwd = mc.pack_plain_weights(w, 0.1, True)
    out = torch.empty(B, cout, H, H, device="cuda"); gx = torch.empty_like(x)
    t_f = bench(lambda: _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), None, B, cin, cout, H, H, 1, 0, None, None, 0, None, None, 0.2, 1.0))
    t_d = bench(lambda: _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), None, _lib.ptr(g), _lib.ptr(wp_bwd), None, None, B, cin, cout, H, H, 1))
    w2 = (w.view(cout, cin) * 0.1).contiguous(); w2t = w2.t().contiguous()
    xv, gv = x.view(B, cin, H * H), g.view(B, cout, H * H)
    o2 = torch.empty(B, cout, H * H, device="cuda"); gx2 = torch.empty(B, cin, H * H, device="cuda")
    t_fb = bench(lambda: torch.matmul(w2, xv, out=o2))
    torch.matmul(w2, xv, out=o2)
    t_db = bench(lambda: torch.matmul(w2t, gv, out=gx2))
    ap_f, ap_b = mc.pack_gemm1x1(w.view(cout, cin), 0.1, False), mc.pack_gemm1x1(w.view(cout, cin), 0.1, True)
    o3 = torch.empty(B, cout, H * H, device="cuda"); gx3 = torch.empty(B, cin, H * H, device="cuda")
    t_f1 = bench(lambda: _lib.call("cagc_gemm1x1", _lib.ptr(o3), _lib.ptr(x), _lib.ptr(ap_f), None, B, cin, cout, H * H, 1.0, 0.0))
    t_d1 = bench(lambda: _lib.call("cagc_gemm1x1", _lib.ptr(gx3), _lib.ptr(g), _lib.ptr(ap_b), None, B, cout, cin, H * H, 1.0, 0.0))
    fl = 2.0 * B * cin * cout * H * H
    e1 = (o3 - o2).abs().max().item() / o2.abs().max().item()
    print(f"{cin:3d}->{cout:3d} @{H:3d}^2 B{B}: cagc_gemm1x1 fwd {t_f1*1e6:7.1f} us ({fl/t_f1/1e12:5.1f} TF)  dgrad {t_d1*1e6:7.1f} us ({fl/t_d1/1e12:5.1f} TF)  rel diff vs rocBLAS {e1:.1e}")
    err = (o2.view_as(out) - out).abs().max().item() / out.abs().max().item()
    print(f"{cin:3d}->{cout:3d} @{H:3d}^2 B{B}: fwd cagc {t_f*1e6:7.1f} us ({fl/t_f/1e12:5.1f} TF)  rocBLAS {t_fb*1e6:7.1f} us ({fl/t_fb/1e12:5.1f} TF) | "
          f"dgrad cagc {t_d*1e6:7.1f} us ({fl/t_d/1e12:5.1f} TF)  rocBLAS {t_db*1e6:7.1f} us ({fl/t_db/1e12:5.1f} TF)  rel diff {err:.1e}")
