"""-m gpu: the launch-count kernels of round 4 (DESIGN §5 "small per-GPU batch") through the C ABI against float64 / against the
per-layer entry points they replace:

* mapping-network layer `cagc_maplin_fwd / _bwd` (reference model.py:137-166 + op/fused_act.py:104-119);
* device-side style mixing `cagc_mix_latent_fwd / _bwd` (model.py:586-594);
* `cagc_modconv_prep_bank` == `cagc_modconv_prep_all` per layer (bit for bit), `cagc_demod_bank` == `cagc_demod_fwd`;
* `cagc_styled_bwd_tail` == `cagc_styled_bwd_finish` + `cagc_demod_bwd` (model.py:249-253 backward);
* a generator forward / backward with the banks on == the same with every layer preparing itself."""
import math

import pytest
import torch

import cagc.model as M
from cagc import _lib
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max() / b.detach().double().abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("R,D,O,lr_mul,act", [(2, 512, 512, 0.01, True), (4, 512, 512, 0.01, True), (32, 512, 512, 0.01, True),
                                             (130, 512, 512, 1.0, True), (3, 512, 40, 0.5, True), (16, 8192, 512, 1.0, True),
                                             (2, 8192, 512, 1.0, True), (16, 512, 1, 1.0, False), (5, 1024, 77, 1.0, False)])
def test_few_row_equal_linear_forward_backward_vs_float64(R, D, O, lr_mul, act):
    """Mapping-network layers (512 -> 512, lr_mul 0.01, fused lrelu) and the discriminator's final linears (8192 -> 512 with
    activation, 512 -> 1 without) on cagc_maplin_fwd / _bwd: output and all three gradients vs float64; the frozen form (input
    gradient only); and the differentiable backward that create_graph=True asks for (R1 through D's final linears)."""
    torch.manual_seed(51)
    lin = M.EqualLinear(D, O, lr_mul=lr_mul, activation="fused_lrelu" if act else None)
    with torch.no_grad():
        lin.bias.copy_(torch.randn(O))
    x = torch.randn(R, D)
    gy = torch.randn(R, O)
    x64 = x.double().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    pre = x64 @ (w64 * lin.scale).t() + b64 * lr_mul
    ling = lin.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    assert mc.map_linear_ok(xg, ling)
    yg = ling(xg)
    assert type(yg.grad_fn).__name__ == "_MapLinearBackward"
    if act:
        gate = (yg.detach() > 0).cpu()
        dis = gate != (pre.detach() > 0)
        assert int(dis.sum()) == 0 or float(pre.detach()[dis].abs().max()) < 1e-5 * float(pre.detach().abs().max())
        y64 = torch.where(gate, pre, 0.2 * pre) * math.sqrt(2)
    else:
        y64 = pre
    g64 = torch.autograd.grad(y64, [x64, w64, b64], gy.double(), retain_graph=True)
    gg = torch.autograd.grad(yg, [xg, ling.weight, ling.bias], gy.to(DEV), retain_graph=True)
    assert _rel(yg, y64) <= 5e-6
    for nm, a, b in zip(("x", "weight", "bias"), gg, g64):
        assert _rel(a, b) <= 5e-6, (nm, _rel(a, b))
    # second order: d/d(weight) of |d y / d x . v|^2  (the R1 pattern) — the backward built from differentiable ops
    v = torch.randn(R, O)
    (gx_c,) = torch.autograd.grad(yg, xg, v.to(DEV), create_graph=True)
    (gw2,) = torch.autograd.grad(gx_c.pow(2).sum(), ling.weight)
    (gx64_c,) = torch.autograd.grad(y64, x64, v.double(), create_graph=True)
    (gw2_64,) = torch.autograd.grad(gx64_c.pow(2).sum(), w64)
    assert _rel(gw2, gw2_64) <= 2e-5, ("second order", _rel(gw2, gw2_64))
    # a frozen layer (teacher / D on the generator step): same kernel, input gradient only
    for p in ling.parameters():
        p.requires_grad_(False)
    xg2 = x.to(DEV).requires_grad_(True)
    y2 = ling(xg2)
    assert torch.equal(y2, yg)
    (gx2,) = torch.autograd.grad(y2, xg2, gy.to(DEV))
    assert torch.equal(gx2, gg[0])


@pytest.mark.parametrize("inj", [1, 5, 13, 14])
def test_mix_latent_device_index(inj):
    torch.manual_seed(52)
    B, n, D = 3, 14, 512
    w0 = torch.randn(B, D, device=DEV, requires_grad=True)
    w1 = torch.randn(B, D, device=DEV, requires_grad=True)
    idx = torch.full((1,), inj, device=DEV, dtype=torch.long)
    assert mc.mix_latent_ok(w0, w1, idx)
    lat = mc._MixLatent.apply(w0, w1, idx, n)
    ref = torch.cat([w0.unsqueeze(1).repeat(1, inj, 1), w1.unsqueeze(1).repeat(1, n - inj, 1)], 1)
    assert torch.equal(lat, ref)
    g = torch.randn(B, n, D, device=DEV)
    ga = torch.autograd.grad(lat, [w0, w1], g)
    gb = torch.autograd.grad(ref, [w0, w1], g)
    for a, b in zip(ga, gb):
        assert _rel(a, b) <= 1e-6


def _layers():
    torch.manual_seed(53)
    shapes = [(154, 154, 3), (154, 77, 3), (77, 39, 3), (39, 39, 3), (512, 512, 3), (39, 3, 1)]
    return [(torch.randn(1, co, ci, k, k, device=DEV), co, ci, k) for ci, co, k in shapes]


def test_prep_bank_equals_prep_all_and_demod_bank_equals_demod_fwd():
    layers = _layers()
    B = 3
    new = lambda n: torch.empty(int(n), device=DEV)
    singles, banked, jobs = [], [], []
    for w, co, ci, k in layers:
        scale = 1.0 / math.sqrt(ci * k * k)
        wino = k == 3
        singles.append(mc.prep_all(w, True, wino, wino))
        outs = [new(_lib.query("cagc_modconv_packed_elems", ci, co, k)), new(_lib.query("cagc_modconv_packed_elems", co, ci, k)),
                torch.empty(co, ci, device=DEV),
                new(_lib.query("cagc_wino_packed_elems", ci, co)) if wino else None,
                new(_lib.query("cagc_wino_packed_elems", co, ci)) if wino else None]
        banked.append(outs)
        jobs.append(_lib.PrepJob(_lib.ptr(w.contiguous()), *[_lib.ptr(t) for t in outs], co, ci, k, scale))
    # more jobs than one launch holds (16): the entry point chunks
    reps = 3
    _lib.call("cagc_modconv_prep_bank", (_lib.PrepJob * (len(jobs) * reps))(*(jobs * reps)), len(jobs) * reps)
    torch.cuda.synchronize()
    for s_, b_ in zip(singles, banked):
        for a, b in zip(s_, b_):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
    # demodulation bank
    djobs, refs, ds = [], [], []
    for (w, co, ci, k), outs in zip(layers, banked):
        s = torch.randn(B, ci, device=DEV)
        d_ref = torch.empty(B, co, device=DEV)
        _lib.call("cagc_demod_fwd", _lib.ptr(d_ref), _lib.ptr(s), _lib.ptr(outs[2]), B, ci, co)
        d = torch.empty(B, co, device=DEV)
        djobs.append(_lib.DemodJob(_lib.ptr(d), _lib.ptr(s), _lib.ptr(outs[2]), ci, co))
        refs.append((d_ref, s))
        ds.append(d)
    _lib.call("cagc_demod_bank", (_lib.DemodJob * len(djobs))(*djobs), len(djobs), B)
    torch.cuda.synchronize()
    for d, (d_ref, _) in zip(ds, refs):
        assert torch.equal(d, d_ref)


@pytest.mark.parametrize("cfg", [(2, 154, 154, True), (3, 77, 39, True), (16, 512, 512, False), (1, 39, 39, True)])
def test_styled_bwd_tail_equals_finish_plus_demod_bwd(cfg):
    B, cin, cout, has_noise = cfg
    torch.manual_seed(54)
    red = torch.randn(3, B, cout, device=DEV)
    bias, nw = torch.randn(cout, device=DEV), torch.randn(1, device=DEV)
    d = torch.rand(B, cout, device=DEV) + 0.5
    s = torch.randn(B, cin, device=DEV)
    wsq = torch.rand(cout, cin, device=DEV)
    # old: finish (gbias, gnw, gd, zero gs) + demod_bwd (gs +=, gwsq)
    gb0, gn0, gd0 = torch.empty(cout, device=DEV), torch.empty(1, device=DEV), torch.empty(B, cout, device=DEV)
    gs0, gw0 = torch.full((B, cin), 7.0, device=DEV), torch.empty(cout, cin, device=DEV)
    _lib.call("cagc_styled_bwd_finish", _lib.ptr(gb0), _lib.ptr(gn0) if has_noise else None, _lib.ptr(gd0), _lib.ptr(gs0), gs0.numel(),
              _lib.ptr(red), _lib.ptr(bias), _lib.ptr(nw) if has_noise else None, _lib.ptr(d), B, cout, 1 if has_noise else 0)
    _lib.call("cagc_demod_bwd", _lib.ptr(gs0), _lib.ptr(gw0), _lib.ptr(gd0), _lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq), B, cin, cout)
    gb1, gn1 = torch.empty(cout, device=DEV), torch.empty(1, device=DEV)
    gs1, gw1 = torch.full((B, cin), -3.0, device=DEV), torch.empty(cout, cin, device=DEV)
    _lib.call("cagc_styled_bwd_tail", _lib.ptr(gb1), _lib.ptr(gn1) if has_noise else None, _lib.ptr(gs1), _lib.ptr(gw1), _lib.ptr(red),
              _lib.ptr(bias), _lib.ptr(nw) if has_noise else None, _lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq), B, cin, cout, 1 if has_noise else 0)
    torch.cuda.synchronize()
    assert torch.equal(gb0, gb1)
    if has_noise:
        assert _rel(gn1, gn0) <= 1e-6
    assert _rel(gs1, gs0) <= 2e-6 and _rel(gw1, gw0) <= 2e-6
    # float64 statement of the same formulas
    r0, r1, r2 = red.double().cpu()
    nwv = float(nw) if has_noise else 0.0
    t = -0.5 * (r2 - bias.double().cpu() * r0 - nwv * r1) * d.double().cpu() ** 2
    gs64 = 2 * s.double().cpu() * (t @ wsq.double().cpu())
    gw64 = t.t() @ (s.double().cpu() ** 2)
    assert _rel(gs1, gs64) <= 5e-6 and _rel(gw1, gw64) <= 5e-6
    # only the weight-side outputs wanted (frozen modulation): gs null
    gw2 = torch.empty(cout, cin, device=DEV)
    _lib.call("cagc_styled_bwd_tail", _lib.ptr(gb1), None, None, _lib.ptr(gw2), _lib.ptr(red), _lib.ptr(bias), _lib.ptr(nw) if has_noise else None,
              _lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq), B, cin, cout, 1 if has_noise else 0)
    assert torch.equal(gw2, gw1)


def test_generator_with_banks_equals_layers_preparing_themselves(monkeypatch):
    """Student-shaped tiny generator (style_dim 512 so that the modulation / preparation banks and the mapping kernels engage):
    image and every gradient with the banks == the same with `_bank_prepare` off (each layer runs cagc_modconv_prep_all +
    cagc_demod_fwd itself) — the same kernels on the same values, so agreement is at summation-order level."""
    torch.manual_seed(55)
    net = M.Generator(32, 512, 2, generator_net_shape=[40, 40, 24, 24, 20, 20, 12, 12]).to(DEV)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    z = [torch.randn(3, 512, device=DEV), torch.randn(3, 512, device=DEV)]
    inj = torch.full((1,), 4, device=DEV, dtype=torch.long)

    def run():
        img = net(z, inject_index=inj, randomize_noise=False)
        gs = torch.autograd.grad(img.abs().mean(), list(net.parameters()), allow_unused=True)
        return img.detach(), gs

    calls = []
    orig = M.Generator._bank_prepare
    monkeypatch.setattr(M.Generator, "_bank_prepare", lambda self, latent, bank: calls.append(1) or orig(self, latent, bank))
    img_a, g_a = run()
    assert calls, "the preparation bank did not engage"
    monkeypatch.setattr(M.Generator, "_bank_prepare", lambda self, latent, bank: None)
    img_b, g_b = run()
    assert _rel(img_a, img_b) <= 1e-6
    for (n, _), a, b in zip(net.named_parameters(), g_a, g_b):
        if b is None:
            assert a is None
            continue
        assert _rel(a, b) <= (2e-5 if b.numel() > 1 else 1e-3), (n, _rel(a, b))
    # fresh noise: ONE flat draw per generator, viewed per layer — two calls differ, statistics of a layer's map are N(0,1)
    with torch.no_grad():
        i1, i2 = net(z, inject_index=inj), net(z, inject_index=inj)
    assert torch.isfinite(i1).all() and not torch.equal(i1, i2)
