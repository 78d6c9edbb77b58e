import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
import torch.nn.functional as F
torch.manual_seed(0)
for (B, Cin, Cout, H, W) in [(2,128,128,64,64),(1,512,512,32,64),(3,20,36,32,32),(1,8,16,32,32),(1,8,16,8,32),(1,16,16,8,32),(1,24,16,8,32),(2,8,16,16,64),(1,11,7,32,32)]:
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda")
    up = mc.pack_wino(w, 1.0, False); out = torch.full((B, Cout, H, W), float("nan"), device="cuda")
    errs = []
    for rep in range(3):
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, Cin, Cout, H, W, 0, None, None, 0, None, None, 0.2, 1.0)
        ref = F.conv2d(x, w, padding=1)
        errs.append(((out - ref).abs().max() / ref.abs().max()).item())
    bad = (out - ref).abs() > 1e-3 * ref.abs().max()
    print((B, Cin, Cout, H, W), ["%.2e" % e for e in errs], "bad frac %.4f" % bad.float().mean().item(),
          "bad rows", sorted(set(bad.nonzero()[:, 2].tolist()))[:12], "bad ch", sorted(set(bad.nonzero()[:, 1].tolist()))[:8])
