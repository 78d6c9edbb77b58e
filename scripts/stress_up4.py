"""Stress of the stream-K hand-off of csrc/conv_up4.hip: the same launch many times (bit-compared with a float64-checked first result),
alternating workgroup shapes / modes so that consecutive launches reuse the slab with different layouts.   python scripts/stress_up4.py"""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ["LIB"])
from cagc.op import modconv as mc
from torch.nn import functional as F
N = int(os.environ.get("N", "150"))
torch.manual_seed(1)
cases = []
for (B, cin, cout, H) in [(2, 128, 256, 64), (6, 64, 64, 254), (3, 64, 40, 10)]:
    hb = H + 1; pitch = (hb + 3) // 4 * 4; ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    _, wp_bwd = mc.pack_plain_weights(w, 0.05, True)
    g = torch.randn(B, cout, ho, ho, device="cuda")
    ref = F.conv_transpose2d(g.double(), w.double() * 0.05, stride=2)
    cases.append(("s2dgrad", (B, cin, cout, H), (wp_bwd, g, hb, pitch), ref))
for (B, cin, cout, H) in [(2, 256, 128, 32), (16, 128, 256, 40)]:
    wt = torch.randn(1, cout, cin, 3, 3, device="cuda")
    wp_fwd, _, _ = mc.pack_weights(wt, True)
    x, s = torch.randn(B, cin, H, H, device="cuda"), torch.rand(B, cin, device="cuda") + 0.5
    cases.append(("upfwd", (B, cin, cout, H), (wp_fwd, x, s), None))
bad = {}
first = {}
for it in range(N):
    for ci, (kind, shape, args, ref) in enumerate(cases):
        for nb in (2, 4):
            B, cin, cout, H = shape
            with _lib.tuning(up4=1, up4_min_ksteps=1, up4_lmin=2 + 2 * (it % 3), up4_nb=nb):
                if kind == "s2dgrad":
                    wp, g, hb, pitch = args
                    out = torch.full((B, cin, hb, pitch), float("nan"), device="cuda")
                    _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(out), _lib.ptr(g), _lib.ptr(wp), B, cin, cout, hb, hb, pitch)
                    v = out[..., :hb]
                    err = float((v.double() - ref).abs().max() / ref.abs().max())
                    ok = err < 1e-5
                else:
                    wp, x, s = args
                    P = _lib.query("cagc_phase_pitch", H)
                    out = torch.full((B, cout, 4, H + 1, P), float("nan"), device="cuda")
                    _lib.call("cagc_modconv_up_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp), _lib.ptr(s), B, cin, cout, H, H)
                    v = out[..., :H + 1].clone()
                    key = (ci, it % 3)      # lmin changes the K split, i.e. the summation order
                    ok = True
                    if (key, nb) in first: ok = torch.equal(v, first[(key, nb)])
                    else: first[(key, nb)] = v
            if not ok:
                bad[(kind, shape, nb)] = bad.get((kind, shape, nb), 0) + 1
torch.cuda.synchronize()
print("launches per case:", N, " bad:", bad if bad else "none", " error word", _lib.get_tuning("up4_error"))
