"""Run ON THE GPU BOX: python scripts/cmp_wino_builds.py libA.so libB.so — the F(4x4) kernel of two builds on the same seeded inputs (linear,
styled and gated-data-gradient launches, both workgroup shapes, REPS launches each): outputs must be BIT-IDENTICAL between the builds and
between repetitions when a change only moves synchronisation / scheduling (a read of a stale LDS buffer changes bits)."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 3 and sys.argv[1] != "--child":
    outs = []
    for lib in sys.argv[1:3]:
        f = f"/tmp/cmp_{lib}.pt"
        r = subprocess.run([sys.executable, __file__, "--child", lib, f], capture_output=True, text=True)
        print(r.stdout.strip()[-400:]); 
        if r.returncode: print(r.stderr[-600:]); sys.exit(1)
        outs.append(torch.load(f))
    bad = 0
    for k in outs[0]:
        same = torch.equal(outs[0][k], outs[1][k])
        bad += 0 if same else 1
        print(f"{k}: {'bit-identical' if same else 'DIFFERENT  max |d| = %.3e' % float((outs[0][k] - outs[1][k]).abs().max())}")
    print("BUILDS", "AGREE" if bad == 0 else f"DIFFER ({bad})")
    sys.exit(1 if bad else 0)
lib, path = sys.argv[2], sys.argv[3]
os.environ["CAGC_WINO4_MIN_WGS"] = "0"
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
_lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", lib)
from cagc.op import modconv as mc
REPS = int(os.environ.get("REPS", "40"))
res, unstable = {}, 0
for hv in (2, 1):
    with _lib.tuning(wino4_hv=hv):
        for (B, cin, cout, H, W) in [(16, 512, 512, 64, 64), (16, 128, 128, 256, 256), (8, 256, 256, 128, 128), (3, 136, 128, 16, 32), (2, 512, 512, 32, 32)]:
            torch.manual_seed(1000 + cin + H)
            x = torch.randn(B, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda")
            s = torch.rand(B, cin, device="cuda") + 0.5; d = torch.rand(B, cout, device="cuda") + 0.5
            noise = torch.randn(B, 1, H, W, device="cuda"); nw = torch.tensor([0.3], device="cuda"); bias = 0.1 * torch.randn(cout, device="cuda")
            up = mc.pack_wino(w, 0.02, False); upb = mc.pack_wino(w, 0.02, True)
            gout = torch.randn(B, cout, H, W, device="cuda"); act = torch.randn(B, cout, H, W, device="cuda"); resd = torch.randn(B, cin, H, W, device="cuda")
            out = torch.empty(B, cout, H, W, device="cuda"); gx = torch.empty(B, cin, H, W, device="cuda")
            def lin(): _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), _lib.ptr(s), B, cin, cout, H, W, 0, _lib.ptr(d), None, 0, None, None, 0.2, 1.0); return out
            def sty(): _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), _lib.ptr(s), B, cin, cout, H, W, 1, _lib.ptr(d), _lib.ptr(noise), B, _lib.ptr(nw), _lib.ptr(bias), 0.2, 2 ** 0.5); return out
            def gat(): _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gout), _lib.ptr(act), _lib.ptr(upb), _lib.ptr(resd), B, cin, cout, H, W, 0.2, 2 ** 0.5); return gx
            for name, f in (("linear", lin), ("styled", sty), ("gated", gat)):
                first = None
                for i in range(REPS):
                    o = f()
                    if first is None: first = o.clone()
                    elif not torch.equal(o, first): unstable += 1
                res[f"hv{hv} {name} B{B} {cin}->{cout} {H}x{W}"] = first.cpu()
torch.save(res, path)
print(f"{lib}: {len(res)} cases x {REPS} launches, {unstable} repetitions differed from the first")
sys.exit(1 if unstable else 0)
