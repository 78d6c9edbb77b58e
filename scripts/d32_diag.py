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
