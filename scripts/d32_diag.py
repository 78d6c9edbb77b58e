"""Localise the Discriminator(32) input-gradient deviation: product on the GPU (fused kernels) vs the same product module
in float64 on the CPU (composed ops), block by block, under CAGC_WINO_DGRAD=1/0.  Run on the GPU box."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F

import cagc.model as M
from cagc.op import modconv as mc
from oracle import ref_model
from _util import load_json, load_npz, rel_err

g = load_npz("discriminator32")
f64 = load_npz("float64_refs")
d = M.Discriminator(32)
d.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["seed"]), strict=True)
d64 = copy.deepcopy(d).double()


def run(model, x):
    acts, grads = [], []
    h = x
    feats = []
    for blk in model.convs:
        h = blk(h)
        h.retain_grad()
        feats.append(h)
    out = model.convs[-1:]  # noqa
    b, c, hh, ww = h.shape
    group = min(b, 4)
    sd = h.view(group, -1, 1, c, hh, ww)
    sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdims=True).squeeze(2).repeat(group, 1, hh, ww)
    o = model.final_conv(torch.cat([h, sd], 1))
    o.retain_grad()
    y = model.final_linear(o.view(b, -1))
    F.softplus(-y).mean().backward()
    return y, feats + [o]


x64 = g["x"].double().requires_grad_(True)
y64, f64s = run(d64, x64)
print("float64 product-CPU vs float64 reference golden: y", rel_err(y64, f64["d32/y"]), "gx", rel_err(x64.grad, f64["d32/gx"]))
for wd in (True, False):
    mc.WINO_DGRAD = wd
    dg = copy.deepcopy(d).cuda()
    xg = g["x"].cuda().requires_grad_(True)
    yg, fg = run(dg, xg)
    print(f"WINO_DGRAD={int(wd)}: y err {rel_err(yg, y64):.3e}  gx err {rel_err(xg.grad, x64.grad):.3e}  (vs fp32 golden gx {rel_err(xg.grad, g['gx']):.3e})")
    for i, (a, b) in enumerate(zip(fg, f64s)):
        print(f"   block {i}: act err {rel_err(a, b):.3e}  grad err {rel_err(a.grad, b.grad):.3e}  shape {tuple(a.shape)}")

# ---- is the block-0 gradient deviation a LeakyReLU gate flip?  compare gates of block 1's two activations, and the sparsity
# of the gradient error
mc.WINO_DGRAD = True
dg = copy.deepcopy(d).cuda()
blk_g, blk_c = dg.convs[1], d64.convs[1]
h0g = dg.convs[0](g["x"].cuda()).detach().requires_grad_(True)
h0c = d64.convs[0](g["x"].double()).detach().requires_grad_(True)
a_g, a_c = blk_g.conv1(h0g), blk_c.conv1(h0c)
b_g, b_c = blk_g.conv2(a_g), blk_c.conv2(a_c)
for nm, pg, pc in (("conv1 out", a_g, a_c), ("conv2 out", b_g, b_c)):
    flips = (pg.detach().cpu() > 0) != (pc.detach() > 0)
    print(f"{nm}: gate disagreements {int(flips.sum())} of {flips.numel()};  |float64 value| at them:",
          [f"{v:.2e}" for v in pc.detach()[flips].abs().tolist()[:8]], " (activation scale", f"{float(pc.abs().max()):.2f})")
yg = blk_g(h0g); yc = blk_c(h0c)
u = torch.randn(yc.shape, dtype=torch.float64)
(gg,) = torch.autograd.grad(yg, h0g, u.float().cuda())
(gc,) = torch.autograd.grad(yc, h0c, u)
err = (gg.double().cpu() - gc).abs()
thr = 1e-5 * float(gc.abs().max())
bad = err > thr
print(f"ResBlock input-gradient: rel err {rel_err(gg, gc):.3e}; elements above 1e-5 of max: {int(bad.sum())} of {bad.numel()}"
      f" ({int(bad.any(1).sum())} distinct (b,y,x) positions)")
if bad.any():
    idx = bad.any(1).nonzero()
    print("   positions (b,y,x):", idx[:12].tolist())
