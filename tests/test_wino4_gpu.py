"""Winograd F(4x4,3x3) kernel (csrc/conv_wino4.hip) through the C ABI against float64 convolutions: linear and styled epilogues
(modulation, demodulation, noise, bias, LeakyReLU), the gated data gradient with residual, K tails (input channels not a
multiple of 8), several channel tiles, single- and multi-tile images.  Bar: 5e-5 of the output scale (observed 0.5-2.2e-5; the
F(2x2) kernel it replaces on these layers holds 5e-6 — the larger transform constants cost ~1.5 digits, the parity bar is 1e-3).
Replaces cuDNN at reference model.py:282 (teacher) and :120 (discriminator conv1 / its data gradient)."""
import os

import pytest
import torch
from torch.nn import functional as F

from cagc import _lib
from cagc.op import modconv as mc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("CAGC_WINO_F4", "1") == "0", reason="F(4x4) disabled")]
DEV = "cuda"
BAR = 5e-5


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


# (B, cin, cout, H, W): channel counts that are multiples of 128 (teacher / discriminator) and ragged ones (the pruned student's
# 154; 200 = 3 tiles + 8 channels), K tails, one and several pixel blocks per image
SHAPES = [(1, 128, 128, 8, 32), (2, 136, 128, 16, 32), (3, 200, 256, 8, 64), (2, 256, 384, 24, 32), (1, 512, 512, 32, 32),
          (2, 128, 128, 64, 96), (2, 154, 154, 16, 32), (1, 160, 200, 8, 64)]


@pytest.fixture(params=[0, 1, 2], ids=["hv_auto", "hv1_64ch_4waves", "hv2_128ch_8waves"])
def hv(request):
    """Workgroup shape of k_wino4 pinned through cagc_set_tuning("wino4_hv"): 64 channels x 4 waves (two workgroups per CU) or
    128 channels x 8 waves (taken only where Cout % 128 == 0; other layers keep the 64-channel shape)."""
    # wino4_min_wgs = 0: these small launches would otherwise take the layer's F(2x2) packing
    with _lib.tuning(wino4_hv=request.param, wino4_min_wgs=0):
        yield request.param


@pytest.mark.parametrize("shape", SHAPES)
def test_wino4_forward_linear_and_styled(shape, hv):
    B, cin, cout, H, W = shape
    kp = (cin + 15) // 16 * 16
    n4 = ((cout + 63) // 64) * 36 * kp * 64          # F(4x4) part: 64-channel tiles x 36 positions
    assert _lib.query("cagc_wino_packed_elems", cin, cout) > n4 and (_lib.query("cagc_wino_packed_elems", cin, cout) - n4) % (16 * kp * 64) == 0   # + F(2x2) part
    torch.manual_seed(41)
    x, w = torch.randn(B, cin, H, W), torch.randn(cout, cin, 3, 3)
    s, d = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    noise, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), 0.1 * torch.randn(cout)
    scale = 1.0 / (cin * 9) ** 0.5
    xg, sg, dg, ng, nwg, bg = (t.to(DEV) for t in (x, s, d, noise, nw, bias))
    up = mc.pack_wino(w.to(DEV), scale, False)
    plain = F.conv2d(x.double(), w.double() * scale, padding=1)
    out = torch.full((B, cout, H, W), float("nan"), device=DEV)
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), None, B, cin, cout, H, W, 0, None, None, 0, None, None, 0.2, 1.0)
    assert rel(out, plain) <= BAR, ("plain", shape, rel(out, plain))
    lin = F.conv2d(x.double() * s.double()[:, :, None, None], w.double() * scale, padding=1) * d.double()[:, :, None, None]
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 0, _lib.ptr(dg), None, 0,
              None, None, 0.2, 1.0)
    assert rel(out, lin) <= BAR, ("modulated", shape, rel(out, lin))
    for nb, nz in ((B, noise), (1, noise[:1])):        # per-sample and shared noise maps
        nzg = nz.contiguous().to(DEV)
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 1, _lib.ptr(dg),
                  _lib.ptr(nzg), nb, _lib.ptr(nwg), _lib.ptr(bg), 0.2, 2 ** 0.5)
        ref = F.leaky_relu(lin + 0.3 * nz.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
        assert rel(out, ref) <= BAR, ("styled", nb, shape, rel(out, ref))


@pytest.mark.parametrize("shape", SHAPES)
def test_wino4_gated_data_gradient(shape, hv):
    """gx = conv_transpose(gout * lrelu'(act_out), W) + residual in one launch (frozen discriminator ConvLayer, model.py:694-716);
    the GEMM's M is the layer's INPUT channel count here."""
    B, cout, cin, H, W = shape          # roles swapped: the GEMM's M = cin takes the shapes' channel-tile cases
    torch.manual_seed(42)
    w = torch.randn(cout, cin, 3, 3)
    scale = 1.0 / (cin * 9) ** 0.5
    gout, act, res = torch.randn(B, cout, H, W), torch.randn(B, cout, H, W), torch.randn(B, cin, H, W)
    upb = mc.pack_wino(w.to(DEV), scale, True)
    gg, ag, rg = gout.to(DEV), act.to(DEV), res.to(DEV)
    gin = gout.double() * torch.where(act > 0, 1.0, 0.2).double() * 2 ** 0.5
    ref = F.conv_transpose2d(gin, w.double() * scale, padding=1)
    gx = torch.full((B, cin, H, W), float("nan"), device=DEV)
    _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(ag), _lib.ptr(upb), None, B, cin, cout, H, W, 0.2, 2 ** 0.5)
    assert rel(gx, ref) <= BAR, ("gated dgrad", shape, rel(gx, ref))
    _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(ag), _lib.ptr(upb), _lib.ptr(rg), B, cin, cout, H, W, 0.2, 2 ** 0.5)
    assert rel(gx, ref + res.double()) <= BAR, ("gated dgrad + residual", shape)


def test_clock_probe_reports_the_shader_clock_and_leaves_results_alone():
    """include/cagc.h cagc_set_clock_probe (bench.py roofline.shader_clock_mhz): every 64th workgroup of every F(4x4) launch
    adds its measured shader clock to a caller-owned accumulator; the output is bit-identical with and without the probe."""
    import ctypes
    B, C, H, W = 2, 128, 32, 64
    torch.manual_seed(7)
    x, w = torch.randn(B, C, H, W, device=DEV), torch.randn(C, C, 3, 3, device=DEV)
    up = mc.pack_wino(w, 0.03, False)
    out0, out1 = torch.empty_like(x), torch.empty_like(x)
    run = lambda o: _lib.call("cagc_wino_conv3x3", _lib.ptr(o), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, W, 0, None, None, 0, None, None, 0.2, 1.0)
    with _lib.tuning(wino4_min_wgs=0):
        run(out0)
        acc = torch.zeros(2, device=DEV)
        lib = _lib.load()
        assert lib.cagc_set_clock_probe(ctypes.c_void_p(acc.data_ptr())) == 0
        try:
            for _ in range(3):
                run(out1)
            torch.cuda.synchronize()
        finally:
            assert lib.cagc_set_clock_probe(None) == 0
        run(out1)                                    # probe off again: the accumulator no longer moves
        torch.cuda.synchronize()
    n, mhz = float(acc[1]), float(acc[0] / acc[1])
    wgs = B * (H // 8) * (W // 32) * (C // 64)       # 64-channel workgroups (a 128-channel shape would be half as many): every 64th samples
    assert n in (3.0 * ((wgs + 63) // 64), 3.0 * ((wgs // 2 + 63) // 64)) and 500.0 < mhz < 3000.0, (n, mhz, wgs)
    assert torch.equal(out0, out1)


# K split of under-filled launches (round 6, prep_device.h wino4_ksplit): K slices of >= 64 channels on 8-wave workgroups, partial
# outputs through a library slab, conv_rd.hip's ordered reduce + the deferred styled epilogue.  (B, cin, cout, H, W, forced ks)
KS_SHAPES = [(2, 512, 512, 32, 32, 8), (2, 512, 512, 32, 32, 2), (1, 256, 384, 24, 32, 4), (2, 136, 128, 16, 32, 2), (1, 528, 256, 8, 32, 8),
             (2, 128, 128, 8, 32, 2)]


@pytest.mark.parametrize("shape", KS_SHAPES)
def test_wino4_ksplit_forward_and_gated_data_gradient(shape):
    B, cin, cout, H, W, ks = shape
    torch.manual_seed(44)
    x, w = torch.randn(B, cin, H, W), torch.randn(cout, cin, 3, 3)
    s, d = torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    noise, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), 0.1 * torch.randn(cout)
    scale = 1.0 / (cin * 9) ** 0.5
    xg, sg, dg, ng, nwg, bg = (t.to(DEV) for t in (x, s, d, noise, nw, bias))
    up = mc.pack_wino(w.to(DEV), scale, False)
    lin = F.conv2d(x.double() * s.double()[:, :, None, None], w.double() * scale, padding=1) * d.double()[:, :, None, None]
    ref = F.leaky_relu(lin + 0.3 * noise.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    plain = F.conv2d(x.double(), w.double() * scale, padding=1)
    outs = {}
    for k in (1, ks):
        with _lib.tuning(wino4_ks=k, wino4_min_wgs=0):
            n0 = _lib.get_tuning("wino4_ks_launches")
            out = torch.full((B, cout, H, W), float("nan"), device=DEV)
            _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 1, _lib.ptr(dg),
                      _lib.ptr(ng), B, _lib.ptr(nwg), _lib.ptr(bg), 0.2, 2 ** 0.5)
            assert rel(out, ref) <= BAR, ("styled", k, shape, rel(out, ref))
            out2 = torch.full((B, cout, H, W), float("nan"), device=DEV)
            _lib.call("cagc_wino_conv3x3", _lib.ptr(out2), _lib.ptr(xg), _lib.ptr(up), None, B, cin, cout, H, W, 0, None, None, 0, None, None, 0.2, 1.0)
            assert rel(out2, plain) <= BAR, ("plain", k, shape, rel(out2, plain))
            n1 = _lib.get_tuning("wino4_ks_launches")
            expect_split = k > 1 and ((cin + 15) // 16 * 16) // 64 >= 2
            assert (n1 - n0 == 2) == expect_split, (k, shape, n1 - n0)
            outs[k] = out.clone()
            out3 = torch.full((B, cout, H, W), float("nan"), device=DEV)      # bit-reproducible: ordered slabs, no atomics
            _lib.call("cagc_wino_conv3x3", _lib.ptr(out3), _lib.ptr(xg), _lib.ptr(up), _lib.ptr(sg), B, cin, cout, H, W, 1, _lib.ptr(dg),
                      _lib.ptr(ng), B, _lib.ptr(nwg), _lib.ptr(bg), 0.2, 2 ** 0.5)
            assert torch.equal(out, out3)
    # gated data gradient (GEMM K = this layer's Cout, M = Cin: needs Cin % 128 == 0 to split)
    if cin % 128 == 0:
        gout, act = torch.randn(B, cout, H, W), torch.randn(B, cout, H, W)
        upb = mc.pack_wino(w.to(DEV), scale, True)
        gin = gout.double() * torch.where(act > 0, 1.0, 0.2).double() * 2 ** 0.5
        gref = F.conv_transpose2d(gin, w.double() * scale, padding=1)
        with _lib.tuning(wino4_ks=ks, wino4_min_wgs=0):
            n0 = _lib.get_tuning("wino4_ks_launches")
            gx = torch.full((B, cin, H, W), float("nan"), device=DEV)
            gg, ag = gout.to(DEV), act.to(DEV)
            _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(ag), _lib.ptr(upb), None, B, cin, cout, H, W, 0.2, 2 ** 0.5)
            assert rel(gx, gref) <= BAR, ("gated dgrad", shape, rel(gx, gref))
            assert _lib.get_tuning("wino4_ks_launches") - n0 == (1 if ((cout + 15) // 16 * 16) // 64 >= 2 else 0)


def test_wino4_ksplit_plan_for_the_small_batch_shapes():
    """per-GPU batch 2 / 8 of configs[1]: the 512-channel layers that take F(4x4) with fewer than 256 128-channel workgroups split K;
    the F(4x4)-or-F(2x2) choice itself (cagc_wino_plan) is what it was"""
    assert _lib.get_tuning("wino4_ks") == 0
    assert _lib.query("cagc_wino_plan", 2, 512, 512, 64, 64) == 4 and _lib.query("cagc_wino_plan", 8, 512, 512, 32, 32) == 4
    assert _lib.query("cagc_wino_plan", 2, 512, 512, 32, 32) == 2 and _lib.query("cagc_wino_plan", 4, 512, 512, 32, 32) == 2
    for B, H, split in ((2, 64, True), (8, 32, True), (4, 64, False), (16, 32, False), (2, 32, False)):
        x, w = torch.randn(B, 512, H, H, device=DEV), torch.randn(512, 512, 3, 3, device=DEV)
        up = mc.pack_wino(w, 0.01, False)
        out = torch.empty_like(x)
        n0 = _lib.get_tuning("wino4_ks_launches")
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, 512, 512, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
        torch.cuda.synchronize()
        assert (_lib.get_tuning("wino4_ks_launches") > n0) == split, (B, H)
