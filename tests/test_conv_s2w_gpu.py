"""Winograd-domain 3x3 stride-2 forward convolution (csrc/conv_s2w.hip: 25 products into 9 sums per 2x2 output tile) against float64,
through the C ABI: `cagc_conv3x3s2_fwd` / `cagc_conv3x3s2_act_fwd` (reference model.py:683-706: Blur(pad=(2,2)) -> EqualConv2d(stride=2,
padding=0), + FusedLeakyReLU).  `cagc_set_tuning("s2w_min_ksteps", 0)` sends the small shapes there; `"s2w_launches"` proves the kernel
took the launch.  The input is the blurred (2Ho+1) x (2Wo+1) plane at a 16-byte row pitch; the pad columns hold NaN here: no valid output
may depend on them.  Coefficients are 0 / +-1: the float64 bar is the direct kernels' 5e-6 loosened to 2e-5."""
import math

import pytest
import torch
from torch.nn import functional as F

from cagc import _lib
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


# (B, cin, cout, Ho, Wo, lmin): everything stream-K (fewer units than workgroups), a whole round + a left-over, K not a multiple of 8
# channels, odd tile rows (Ho odd), non-square planes, a tile row shorter than a wave's 16 tiles
SHAPES = [(2, 128, 256, 32, 32, 8), (1, 512, 512, 16, 16, 8), (3, 20, 64, 4, 4, 2), (16, 64, 64, 8, 8, 4), (5, 24, 128, 20, 20, 2),
          (3, 36, 64, 7, 12, 2), (9, 64, 64, 63, 64, 8), (2, 256, 512, 64, 64, 8),
          (2, 32, 77, 8, 8, 2), (2, 32, 48, 8, 8, 2), (3, 24, 154, 12, 12, 2)]      # 5 / 3 channel blocks per wave, ragged last block


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_stride2_forward_winograd_kernel(shape, act):
    B, cin, cout, ho, wo, lmin = shape
    torch.manual_seed(41)
    hin, win = 2 * ho + 1, 2 * wo + 1
    pitch = (win + 3) // 4 * 4
    w = torch.randn(cout, cin, 3, 3, device=DEV)
    bias = torch.randn(cout, device=DEV)
    scale = 1.0 / math.sqrt(cin * 9)
    wp_fwd, _ = mc.pack_plain_weights(w, scale, True)
    x = torch.full((B, cin, hin, pitch), float("nan"), device=DEV)
    x[..., :win] = torch.randn(B, cin, hin, win, device=DEV)
    ref = F.conv2d(x[..., :win].double(), w.double() * scale, stride=2)
    if act:
        ref = F.leaky_relu(ref + bias.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2.0)
    out = torch.full((B, cout, ho, wo), float("nan"), device=DEV)
    n0 = _lib.get_tuning("s2w_launches")
    with _lib.tuning(s2w=1, s2w_min_ksteps=0, s2w_lmin=lmin):
        if act:
            _lib.call("cagc_conv3x3s2_act_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(bias), B, cin, cout, hin, win, pitch, 0.2, math.sqrt(2.0))
        else:
            _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), B, cin, cout, hin, win, pitch)
    torch.cuda.synchronize()
    assert _lib.get_tuning("s2w_launches") == n0 + 1, "the launch did not reach conv_s2w.hip"
    assert rel(out, ref) <= TOL, (shape, act, rel(out, ref))
    assert _lib.get_tuning("up4_error") == 0


def test_shapes_it_does_not_take_stay_on_the_direct_kernels():
    """7 channel blocks (neither 4 / 5 per wave nor 3) / output rows that are not whole 16-byte stores: same results with the knob on and off"""
    torch.manual_seed(42)
    for (B, cin, cout, ho) in [(2, 77, 100, 8), (2, 64, 64, 5)]:
        hin = 2 * ho + 1
        pitch = (hin + 3) // 4 * 4
        w = torch.randn(cout, cin, 3, 3, device=DEV)
        wp_fwd, _ = mc.pack_plain_weights(w, 0.05, True)
        x = torch.randn(B, cin, hin, pitch, device=DEV)
        outs = []
        for on in (1, 0):
            out = torch.zeros(B, cout, ho, ho, device=DEV)
            with _lib.tuning(s2w=on, s2w_min_ksteps=0):
                _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), B, cin, cout, hin, hin, pitch)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])


def test_stream_k_handoff_is_bit_reproducible():
    torch.manual_seed(43)
    for (B, cin, cout, ho) in [(2, 256, 128, 16), (16, 128, 256, 20)]:
        hin = 2 * ho + 1
        pitch = (hin + 3) // 4 * 4
        w = torch.randn(cout, cin, 3, 3, device=DEV)
        wp_fwd, _ = mc.pack_plain_weights(w, 0.05, True)
        x = torch.randn(B, cin, hin, pitch, device=DEV)
        first = None
        with _lib.tuning(s2w=1, s2w_min_ksteps=0, s2w_lmin=2):
            for it in range(20):
                out = torch.full((B, cout, ho, ho), float("nan"), device=DEV)
                _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), B, cin, cout, hin, hin, pitch)
                if first is None:
                    first = out.clone()
                else:
                    assert torch.equal(out, first), (B, cin, cout, ho, it)
        torch.cuda.synchronize()
    assert _lib.get_tuning("up4_error") == 0


# (B, cin, cout, H, W, lmin): the student's up layers — gx has cin channels (GEMM M: 5 / 5+5 / 3 blocks per wave), K = cout
UPD_SHAPES = [(2, 77, 39, 16, 16, 2), (2, 154, 77, 8, 8, 2), (3, 39, 20, 12, 20, 2), (16, 154, 154, 4, 4, 4), (2, 128, 64, 16, 16, 8),
              (4, 77, 39, 64, 64, 8), (2, 80, 36, 7, 8, 2)]


@pytest.mark.parametrize("with_gs", [True, False])
@pytest.mark.parametrize("shape", UPD_SHAPES)
def test_up_layer_data_gradient_on_the_winograd_kernel(shape, with_gs):
    """`cagc_modconv_up_dgrad` (data gradient of reference model.py:259-270's conv_transpose2d): gx = s * conv2d(gT, W^T, stride 2) on the
    phase-planar gradient, gs += sum_yx (unscaled gx) * x — the PLANAR form of csrc/conv_s2w.hip with 5 / 3 / 4 channel blocks per wave."""
    B, cin, cout, H, W, lmin = shape
    torch.manual_seed(44)
    wt = torch.randn(1, cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    _, wp_bwd, _ = mc.pack_weights(wt.to(DEV), True)
    x, s = torch.randn(B, cin, H, W), torch.rand(B, cin) + 0.5
    xg, sg = x.to(DEV), s.to(DEV)
    P = _lib.query("cagc_phase_pitch", W)
    gfull = torch.randn(B, cout, 2 * H + 1, 2 * W + 1)
    gt = torch.full((B, cout, 4, H + 1, P), float("nan"))
    for py in range(2):
        for px in range(2):
            sub = gfull[:, :, py::2, px::2]
            gt[:, :, py * 2 + px, :, :W + 1] = 0.0
            gt[:, :, py * 2 + px, :sub.shape[2], :sub.shape[3]] = sub
    gtg = gt.to(DEV)
    wd = wt[0].double() * scale
    raw = F.conv2d(gfull.double(), wd.transpose(0, 1), stride=2)
    gx = torch.full((B, cin, H, W), float("nan"), device=DEV)
    gs0 = torch.randn(B, cin)
    gs = gs0.clone().to(DEV)
    n0 = _lib.get_tuning("s2w_launches")
    with _lib.tuning(s2w=1, s2w_min_ksteps=0, s2w_lmin=lmin):
        _lib.call("cagc_modconv_up_dgrad", _lib.ptr(gx), _lib.ptr(gs) if with_gs else None, _lib.ptr(gtg), _lib.ptr(wp_bwd), _lib.ptr(sg),
                  _lib.ptr(xg) if with_gs else None, B, cin, cout, H, W)
    torch.cuda.synchronize()
    assert _lib.get_tuning("s2w_launches") == n0 + 1, "the launch did not reach conv_s2w.hip"
    assert rel(gx, raw * s.double()[:, :, None, None]) <= TOL, (shape, rel(gx, raw * s.double()[:, :, None, None]))
    if with_gs:
        gsref = gs0.double() + (raw * x.double()).sum([2, 3])
        assert rel(gs, gsref) <= 5e-5, (shape, rel(gs, gsref))      # cancelling sums over H*W pixels
    assert _lib.get_tuning("up4_error") == 0

