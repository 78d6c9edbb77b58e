import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def sampled(t, cfg):
    """what tests/golden/generator512.npz keeps of a tensor (oracle/gen_golden.py `sampled`): all of it when small, else a strided sample"""
    flat = t.detach().reshape(-1)
    return flat if flat.numel() < cfg["full_below"] else flat[::cfg["sample_stride"]]


def assert_grad_matches_sample(gr, sample, sums, cfg, tol, what):
    """a gradient against its fixture entry: the kept elements at `tol` of the gradient's max, and — for tensors kept only as a
    sample — the (sum, abs-sum) checksums at 10 x tol of the abs-sum (a sum of n rounded terms)"""
    gmax = float(sums[2])
    got = sampled(gr, cfg).double().cpu()
    err = (got - sample.double()).abs().max().item() / (gmax if gmax > 0 else 1.0)
    assert err <= tol, f"{what}: sampled rel err {err:.3e} > {tol:.1e}"
    if gr.numel() >= cfg["full_below"] and float(sums[1]) > 0:
        e_sum = abs(float(gr.double().sum()) - float(sums[0])) / float(sums[1])
        e_abs = abs(float(gr.double().abs().sum()) - float(sums[1])) / float(sums[1])
        assert max(e_sum, e_abs) <= 10 * tol, f"{what}: checksum rel err {max(e_sum, e_abs):.3e}"


def rel_err(a, b):
    """max|a-b| / max|b| — the parity metric of SURVEY.md §8(d)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def assert_close(a, b, tol, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


# ---------------------------------------------------------------------------------------------------
# LeakyReLU gate analysis.  A random-init net has pre-activations arbitrarily close to 0; where fp32 rounding puts one
# on the other side of 0 than float64 does, the backward pass legitimately differs by (1 - 0.2) * sqrt(2) * g on that
# element's receptive field.  These helpers find such elements and the input positions they can reach, so that tests
# can (a) prove every disagreement is at rounding level and (b) hold everything else to the parity bar.
# ---------------------------------------------------------------------------------------------------
def _activated_modules(model):
    import cagc.model as M
    from cagc.op import FusedLeakyReLU
    for name, m in model.named_modules():
        if isinstance(m, M.ConvLayer) and isinstance(m[-1], (FusedLeakyReLU, M.ScaledLeakyReLU)):
            yield name, m
        elif isinstance(m, M.EqualLinear) and m.activation:
            yield name, m


def forward_with_activations(model, x):
    outs, hooks = {}, []
    for name, m in _activated_modules(model):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: outs.__setitem__(name, out)))
    try:
        y = model(x)
    finally:
        for h in hooks:
            h.remove()
    return y, outs


def gate_flips(outs_a, outs_ref64):
    """[(layer, flat index, |reference value| / max|reference|)] where sign(out) differs between the two runs."""
    flips = []
    for name, r in outs_ref64.items():
        a = outs_a[name].detach().cpu()
        d = ((a > 0) != (r.detach() > 0)).flatten().nonzero().flatten().tolist()
        scale = float(r.detach().abs().max())
        for i in d:
            flips.append((name, i, float(r.detach().flatten()[i].abs()) / scale))
    return flips


def reach_mask(outs_ref64, flips, x64):
    """bool [B,1,H,W]: input positions whose gradient can depend on a flipped gate (support of d out[e] / d x, float64)."""
    mask = torch.zeros(x64.shape[0], 1, x64.shape[2], x64.shape[3], dtype=torch.bool)
    for name, i, _ in flips:
        (g,) = torch.autograd.grad(outs_ref64[name].flatten()[i], x64, retain_graph=True)
        mask |= (g != 0).any(1, keepdim=True)
    return mask
