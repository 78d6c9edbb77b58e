"""wgrad_s2 / wgrad_s1 (op/conv_closure.py) vs float64 torch on the CPU, error per tap.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from cagc.op import conv_closure as cc

torch.manual_seed(0)
for (B, cin, cout, ho, wo) in [(2, 20, 36, 8, 10), (4, 512, 512, 32, 32), (2, 128, 256, 32, 32), (1, 32, 64, 512, 512), (3, 64, 48, 16, 16)]:
    hb, wb = 2 * ho + 1, 2 * wo + 1
    xb = torch.randn(B, cin, hb, wb)
    g = torch.randn(B, cout, ho, wo)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xb.double(), w, stride=2)
    (ref,) = torch.autograd.grad(y, w, g.double())
    got = cc.wgrad_s2(g.cuda(), xb.cuda(), 1.0).double().cpu()
    e = (got - ref).abs()
    print(f"s2 B{B} {cin}->{cout} out {ho}x{wo}: rel err {float(e.max() / ref.abs().max()):.3e}; per tap:",
          [f"{float(e[:, :, t // 3, t % 3].max() / ref.abs().max()):.1e}" for t in range(9)])
    # pitched operand
    pitch = (wb + 3) // 4 * 4
    xp = torch.zeros(B, cin, hb, pitch)
    xp[..., :wb] = xb
    got2 = cc.wgrad_s2(g.cuda(), xp.cuda(), 1.0, in_pitch=pitch).double().cpu()
    print(f"    pitched operand: rel err {float((got2 - ref).abs().max() / ref.abs().max()):.3e}")
for (B, cin, cout, H, W, k) in [(2, 20, 36, 18, 20, 3), (2, 128, 128, 32, 32, 3), (2, 3, 128, 64, 64, 1), (2, 512, 512, 16, 16, 1)]:
    x = torch.randn(B, cin, H, W)
    g = torch.randn(B, cout, H, W)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), w, padding=k // 2), w, g.double())
    got = cc.wgrad_s1(g.cuda(), x.cuda(), k, 1.0).double().cpu()
    print(f"s1 k{k} B{B} {cin}->{cout} {H}x{W}: rel err {float((got - ref).abs().max() / ref.abs().max()):.3e}")
