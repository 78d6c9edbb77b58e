"""Fused-phase stride-2 transposed convolution (csrc/conv_up4.hip) against float64 and against the per-parity register-direct kernel
(csrc/conv_rd.hip) it replaces on large launches, through the C ABI.

`cagc_modconv_up_fwd` (reference model.py:259-270, the conv_transpose2d before the blur) and `cagc_conv3x3s2_dgrad` (data gradient of
model.py:693-706) route a launch to the persistent stream-K kernel when it has at least `up4_min_ksteps` (256 positions x 64 channels)
units; `cagc_set_tuning("up4_min_ksteps", 0)` sends the small test shapes there (`"up4_launches"` proves it): everything stream-K (fewer units than workgroups: a
unit's K range is spread over many jobs), one whole round + a left-over, ragged last position tile, K not a multiple of 8 channels,
non-square planes, with and without the input modulation, in both workgroup shapes (`up4_nb` 2 / 4: 32 / 64 positions per wave) and with
the K rotation of a workgroup's first whole unit (`up4_rotate`).  The stream-K hand-off (slab + agent-scope release / acquire) is exercised
under uneven load from a second stream and must be bit-reproducible."""
import math

import pytest
import torch
from torch.nn import functional as F

from cagc import _lib
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 5e-6    # fp32 MFMA = exact fp32 FMA chain: rounding-level agreement with float64


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def phase_planar(full, H, W):
    B, C = full.shape[:2]
    P = _lib.query("cagc_phase_pitch", W)
    t = torch.zeros(B, C, 4, H + 1, P, dtype=full.dtype)
    for py in range(2):
        for px in range(2):
            sub = full[:, :, py::2, px::2]
            t[:, :, py * 2 + px, :sub.shape[2], :sub.shape[3]] = sub
    return t


def no_spin_timeout():
    assert _lib.get_tuning("up4_error") == 0, "a stream-K owner gave up waiting for a contributor"


# (B, cin, cout, H, W, lmin)
UP_SHAPES = [(2, 512, 256, 32, 32, 8), (1, 256, 128, 64, 64, 8), (3, 64, 64, 12, 20, 2), (16, 128, 512, 8, 8, 8), (5, 24, 64, 9, 9, 4),
             (2, 36, 128, 80, 48, 2), (9, 64, 64, 127, 127, 8)]


@pytest.mark.parametrize("nb", [2, 4])
@pytest.mark.parametrize("modulated", [True, False])
@pytest.mark.parametrize("shape", UP_SHAPES)
def test_up_fwd_fused_phase_kernel(shape, modulated, nb):
    B, cin, cout, H, W, lmin = shape
    torch.manual_seed(21)
    wt = torch.randn(1, cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    wp_fwd, _, _ = mc.pack_weights(wt.to(DEV), True)
    x = torch.randn(B, cin, H, W)
    s = torch.rand(B, cin) + 0.5 if modulated else None
    xg = x.to(DEV)
    sg = s.to(DEV) if modulated else None
    P = _lib.query("cagc_phase_pitch", W)
    wd = wt[0].double() * scale
    xs = x.double() * (s.double()[:, :, None, None] if modulated else 1.0)
    tref = phase_planar(F.conv_transpose2d(xs, wd.transpose(0, 1), stride=2), H, W)
    outs = {}
    for up4 in (1, 0):
        t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
        n0 = _lib.get_tuning("up4_launches")
        with _lib.tuning(up4=up4, up25=0, up4_min_ksteps=0, up4_lmin=lmin, up4_nb=nb):
            _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(xg), _lib.ptr(wp_fwd), _lib.ptr(sg), B, cin, cout, H, W)
        torch.cuda.synchronize()
        assert _lib.get_tuning("up4_launches") == n0 + up4
        assert rel(t[..., :W + 1], tref[..., :W + 1]) <= TOL, (up4, shape, modulated, rel(t[..., :W + 1], tref[..., :W + 1]))
        outs[up4] = t[..., :W + 1]
    no_spin_timeout()
    assert rel(outs[1], outs[0]) <= 2e-6


# (B, cin, cout, H): D's stride-2 conv on the blurred (H+1)^2 input; the data gradient has M = cin, K = cout
S2_SHAPES = [(2, 128, 256, 64, 8), (1, 512, 512, 32, 8), (3, 64, 40, 10, 2), (16, 64, 64, 4, 2), (7, 128, 24, 30, 4), (6, 64, 64, 254, 8)]


@pytest.mark.parametrize("nb", [2, 4])
@pytest.mark.parametrize("shape", S2_SHAPES)
def test_stride2_data_gradient_fused_phase_kernel(shape, nb):
    B, cin, cout, H, lmin = shape
    torch.manual_seed(22)
    hb = H + 1
    pitch = (hb + 3) // 4 * 4
    ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    _, wp_bwd = mc.pack_plain_weights(w.to(DEV), scale, True)
    g = torch.randn(B, cout, ho, ho)
    gg = g.to(DEV)
    gref = F.conv_transpose2d(g.double(), w.double() * scale, stride=2)
    outs = {}
    for up4 in (1, 0):
        gx = torch.full((B, cin, hb, pitch), float("nan"), device=DEV)
        n0 = _lib.get_tuning("up4_launches")
        with _lib.tuning(up4=up4, up25=0, up4_min_ksteps=0, up4_lmin=lmin, up4_nb=nb):
            _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch)
        torch.cuda.synchronize()
        assert _lib.get_tuning("up4_launches") == n0 + up4
        assert rel(gx[..., :hb], gref) <= TOL, (up4, shape, rel(gx[..., :hb], gref))
        outs[up4] = gx[..., :hb]
    no_spin_timeout()
    assert rel(outs[1], outs[0]) <= 2e-6


def test_fused_phase_kernel_declines_what_it_does_not_take():
    """ragged channel tiles (the pruned student's 154 / 77 / 39 channels) stay on conv_rd.hip: same results with the knob on and off"""
    torch.manual_seed(23)
    B, cin, cout, H = 2, 77, 39, 16
    wt = torch.randn(1, cout, cin, 3, 3)
    wp_fwd, _, _ = mc.pack_weights(wt.to(DEV), True)
    x, s = torch.randn(B, cin, H, H, device=DEV), torch.rand(B, cin, device=DEV) + 0.5
    P = _lib.query("cagc_phase_pitch", H)
    outs = []
    for up4 in (1, 0):
        t = torch.zeros(B, cout, 4, H + 1, P, device=DEV)
        with _lib.tuning(up4=up4, up25=0, up4_min_ksteps=0):
            _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H)
        outs.append(t)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("nb,rotate", [(2, 0), (4, 0), (2, 4)])
def test_stream_k_handoff_is_bit_reproducible_under_uneven_load(nb, rotate):
    """The slab hand-off between workgroups (agent-scope release -> flag -> acquire) with a second stream keeping some CUs busy and the
    consumer's L1 warm: 40 launches of a launch that is ALL stream-K (fewer units than workgroups) and of one with a whole round + a
    left-over must give the first launch's bits every time."""
    torch.manual_seed(24)
    side = torch.cuda.Stream()
    junk = torch.randn(64, 1024, 1024, device=DEV)
    for (B, cin, cout, H) in [(2, 256, 128, 32), (16, 128, 256, 40)]:
        wt = torch.randn(1, cout, cin, 3, 3)
        wp_fwd, _, _ = mc.pack_weights(wt.to(DEV), True)
        x, s = torch.randn(B, cin, H, H, device=DEV), torch.rand(B, cin, device=DEV) + 0.5
        P = _lib.query("cagc_phase_pitch", H)
        first = None
        with _lib.tuning(up4=1, up25=0, up4_min_ksteps=0, up4_lmin=2, up4_nb=nb, up4_rotate=rotate):
            for it in range(40):
                t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
                if it % 2:
                    with torch.cuda.stream(side):
                        for _ in range(1 + it % 5):
                            junk.mul_(1.0001)
                _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H)
                v = t[..., :H + 1].clone()
                if first is None:
                    first = v
                else:
                    assert torch.equal(v, first), (B, cin, cout, H, it)
        torch.cuda.synchronize()
    no_spin_timeout()
