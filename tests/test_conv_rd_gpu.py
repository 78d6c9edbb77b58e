"""Register-direct implicit-GEMM kernel (csrc/conv_rd.hip) against float64 convolutions, through the C ABI.

Every entry point of the convolution family routes a launch to the register-direct kernel or the LDS-staged one by a
launch-shape heuristic; `cagc_set_tuning` pins the shape so that each variant — plain 2-D tiles, linearised tiles,
K split across the waves of a workgroup (kw = 2, 4, partial sums through LDS), K split across workgroups (fp32
atomics), channel sub-tiles (mb) and the LDS-staged kernel itself (rd = 0) — computes the same shapes, incl. ragged
channel counts (39 / 77 / 154: K padded to 8, M to the channel tile), odd spatial sizes and multi-image tiles.
Replaces cuDNN at reference model.py:267,276,282 (modulated convs) and :683-706 (D's stride-2 convs)."""
import math

import pytest
import torch
from torch.nn import functional as F

from cagc import _lib
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 5e-6    # fp32 MFMA = exact fp32 FMA chain: rounding-level agreement with float64 (observed 2-7e-7)

CONFIGS = {
    "default": {},
    "lds_kernel": {"rd": 0},
    "rd_everything": {"rd_min_wgs": 1, "rd_atomic_below": 0},
    "kw2": {"rd_min_wgs": 1, "rd_kw": 2, "rd_atomic_below": 0},
    "kw4": {"rd_min_wgs": 1, "rd_kw": 4, "rd_atomic_below": 0},
    "fill_no_atomics": {"rd_min_wgs": 1 << 20, "rd_atomic_below": 0},
    "fill_atomics": {"rd_min_wgs": 1 << 20, "rd_atomic_below": 1 << 20, "rd_split_wgs": 256},
    "kw1_atomics": {"rd_min_wgs": 1, "rd_kw": 1, "rd_atomic_below": 1 << 20, "rd_split_wgs": 512},
    "mb4": {"rd_min_wgs": 1, "rd_mb": 4, "rd_atomic_below": 0},
}
DEFAULTS = {"rd": 1, "rd_min_wgs": 512, "rd_mb": 0, "rd_kw": 0, "rd_split": 1, "rd_atomic_below": 160, "rd_split_wgs": 512}


@pytest.fixture(params=list(CONFIGS))
def tuning(request):
    lib = _lib.load()
    for k, v in {**DEFAULTS, **CONFIGS[request.param]}.items():
        assert lib.cagc_set_tuning(k.encode(), v) == 0
    yield request.param
    for k, v in DEFAULTS.items():
        lib.cagc_set_tuning(k.encode(), v)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def phase_planar(full, H, W):
    """[B,C,2H+1,2W+1] -> [B,C,4,H+1,P] as the transposed-conv kernels lay it out (zeros where a phase has no sample)."""
    B, C = full.shape[:2]
    P = _lib.query("cagc_phase_pitch", W)
    t = torch.zeros(B, C, 4, H + 1, P, dtype=full.dtype)
    for py in range(2):
        for px in range(2):
            sub = full[:, :, py::2, px::2]
            t[:, :, py * 2 + px, :sub.shape[2], :sub.shape[3]] = sub
    return t


# (B, cin, cout, H): D's stride-2 conv, input blurred to (H+1)x(H+1) at a 16-byte row pitch
S2_SHAPES = [(3, 20, 36, 10), (2, 77, 39, 16), (16, 64, 64, 4), (5, 154, 160, 8), (2, 128, 256, 64), (1, 512, 512, 32)]


@pytest.mark.parametrize("shape", S2_SHAPES)
def test_stride2_conv_forward_and_data_gradient(shape, tuning):
    B, cin, cout, H = shape
    torch.manual_seed(11)
    hb = H + 1
    pitch = (hb + 3) // 4 * 4
    ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    wp_fwd, wp_bwd = mc.pack_plain_weights(w.to(DEV), scale, True)
    x = torch.randn(B, cin, hb, pitch)
    g = torch.randn(B, cout, ho, ho)
    xg, gg = x.to(DEV), g.to(DEV)          # device copies stay alive across the launches that read them
    out = torch.full((B, cout, ho, ho), float("nan"), device=DEV)
    _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(wp_fwd), B, cin, cout, hb, hb, pitch)
    ref = F.conv2d(x[..., :hb].double(), w.double() * scale, stride=2)
    assert rel(out, ref) <= TOL, ("fwd", tuning, shape, rel(out, ref))
    gx = torch.full((B, cin, hb, pitch), float("nan"), device=DEV)
    _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch)
    gref = F.conv_transpose2d(g.double(), w.double() * scale, stride=2)
    assert rel(gx[..., :hb], gref) <= TOL, ("dgrad", tuning, shape, rel(gx[..., :hb], gref))


# launches big enough (>= 512 workgroups of 128 channels x 256 pixels) for the vector-operand stride-2 kernel (k_conv_s2v): full tiles;
# a ragged last tile + ragged channel count; rows shorter than a wave's 64 pixels (a wave spans 4 rows, a tile 2 images)
S2V_SHAPES = [(4, 16, 256, 256), (5, 24, 250, 200), (256, 8, 256, 32)]


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("shape", S2V_SHAPES)
def test_stride2_forward_vector_operand_kernel(shape, act):
    B, cin, cout, H = shape
    torch.manual_seed(13)
    hb = H + 1
    pitch = (hb + 3) // 4 * 4
    ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3, device=DEV)
    bias = torch.randn(cout, device=DEV)
    scale = 1.0 / math.sqrt(cin * 9)
    wp_fwd, _ = mc.pack_plain_weights(w, scale, True)
    x = torch.randn(B, cin, hb, pitch, device=DEV)
    ref = F.conv2d(x[..., :hb].double(), w.double() * scale, stride=2)
    if act:
        ref = F.leaky_relu(ref + bias.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2.0)
    outs = {}
    for s2v in (1, 0):     # the general register-direct kernel computes the same launch: both against float64, and against each other
        out = torch.full((B, cout, ho, ho), float("nan"), device=DEV)
        with _lib.tuning(rd_s2v=s2v, s2w=0):
            if act:
                _lib.call("cagc_conv3x3s2_act_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(bias), B, cin, cout, hb, hb, pitch, 0.2, math.sqrt(2.0))
            else:
                _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp_fwd), B, cin, cout, hb, hb, pitch)
        assert rel(out, ref) <= TOL, (s2v, shape, act, rel(out, ref))
        outs[s2v] = out
    assert torch.equal(outs[0], outs[1]), "same fp32 FMA chain per output (K order: channels, then taps row-major)"


# (B, cin, cout, H): modulated transposed conv (phase-planar output) and its data gradient with the fused gs reduction
UP_SHAPES = [(3, 20, 36, 5), (2, 77, 39, 16), (16, 154, 154, 4), (4, 154, 77, 8), (2, 512, 256, 32), (1, 256, 128, 64)]


@pytest.mark.parametrize("shape", UP_SHAPES)
def test_transposed_conv_forward_and_data_gradient(shape, tuning):
    B, cin, cout, H = shape
    torch.manual_seed(12)
    W = H
    wt = torch.randn(1, cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    wp_fwd, wp_bwd, _ = mc.pack_weights(wt.to(DEV), True)
    x, s = torch.randn(B, cin, H, W), torch.rand(B, cin) + 0.5
    xg, sg = x.to(DEV), s.to(DEV)          # device copies stay alive across the launches that read them
    P = _lib.query("cagc_phase_pitch", W)
    t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
    _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(xg), _lib.ptr(wp_fwd), _lib.ptr(sg), B, cin, cout, H, W)
    wd = wt[0].double() * scale                                     # [cout, cin, 3, 3]
    full = F.conv_transpose2d(x.double() * s.double()[:, :, None, None], wd.transpose(0, 1), stride=2)
    tref = phase_planar(full, H, W)
    assert rel(t[..., :W + 1], tref[..., :W + 1]) <= TOL, ("up_fwd", tuning, shape, rel(t[..., :W + 1], tref[..., :W + 1]))
    # data gradient: gx = s * conv2d(gT, W^T, stride 2), gs = sum_p (unscaled gx) * x
    gfull = torch.randn(B, cout, 2 * H + 1, 2 * W + 1)
    gtg = phase_planar(gfull, H, W).to(DEV)
    gx = torch.full((B, cin, H, W), float("nan"), device=DEV)
    gs = torch.zeros(B, cin, device=DEV)
    _lib.call("cagc_modconv_up_dgrad", _lib.ptr(gx), _lib.ptr(gs), _lib.ptr(gtg), _lib.ptr(wp_bwd), _lib.ptr(sg), _lib.ptr(xg),
              B, cin, cout, H, W)
    raw = F.conv2d(gfull.double(), wd.transpose(0, 1), stride=2)    # [B, cin, H, W]
    assert rel(gx, raw * s.double()[:, :, None, None]) <= TOL, ("up_dgrad", tuning, shape)
    gsref = (raw * x.double()).sum([2, 3])
    assert rel(gs, gsref) <= 2e-5, ("up_dgrad gs", tuning, shape, rel(gs, gsref))    # cancelling sums over H*W pixels


# (B, cin, cout, H, W, ksize)
PLAIN_SHAPES = [(3, 20, 36, 6, 10, 3), (16, 154, 154, 4, 4, 3), (2, 77, 39, 16, 16, 3), (2, 512, 512, 8, 8, 3),
                (2, 39, 77, 12, 20, 1), (1, 128, 128, 40, 24, 3)]


@pytest.mark.parametrize("shape", PLAIN_SHAPES)
def test_plain_modulated_conv_forward_styled_epilogue_and_data_gradient(shape, tuning):
    B, cin, cout, H, W, k = shape
    torch.manual_seed(13)
    wt = torch.randn(1, cout, cin, k, k)
    scale = 1.0 / math.sqrt(cin * k * k)
    wp_fwd, wp_bwd, _ = mc.pack_weights(wt.to(DEV), True)
    x, s, d = torch.randn(B, cin, H, W), torch.rand(B, cin) + 0.5, torch.rand(B, cout) + 0.5
    noise, nw, bias = torch.randn(B, 1, H, W), torch.tensor([0.3]), 0.1 * torch.randn(cout)
    xg, sg, dg, ng, nwg, bg = (v.to(DEV) for v in (x, s, d, noise, nw, bias))   # alive across the launches that read them
    wd = wt[0].double() * scale
    lin = F.conv2d(x.double() * s.double()[:, :, None, None], wd, padding=k // 2) * d.double()[:, :, None, None]
    for epi in (0, 1):
        out = torch.full((B, cout, H, W), float("nan"), device=DEV)
        _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(wp_fwd), _lib.ptr(sg), B, cin, cout, H, W, k,
                  epi, _lib.ptr(dg), _lib.ptr(ng) if epi else None, B if epi else 0, _lib.ptr(nwg) if epi else None,
                  _lib.ptr(bg) if epi else None, 0.2, 2 ** 0.5)
        ref = lin
        if epi:
            ref = F.leaky_relu(lin + 0.3 * noise.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
        assert rel(out, ref) <= TOL, ("fwd", epi, tuning, shape, rel(out, ref))
    gz = torch.randn(B, cout, H, W)
    gzg = gz.to(DEV)
    gx = torch.full((B, cin, H, W), float("nan"), device=DEV)
    gs = torch.zeros(B, cin, device=DEV)
    _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), _lib.ptr(gs), _lib.ptr(gzg), _lib.ptr(wp_bwd), _lib.ptr(sg), _lib.ptr(xg),
              B, cin, cout, H, W, k)
    raw = F.conv_transpose2d(gz.double(), wd, padding=k // 2)
    assert rel(gx, raw * s.double()[:, :, None, None]) <= TOL, ("dgrad", tuning, shape)
    assert rel(gs, (raw * x.double()).sum([2, 3])) <= 2e-5, ("dgrad gs", tuning, shape)


def test_set_tuning_rejects_unknown_keys():
    lib = _lib.load()
    assert lib.cagc_set_tuning(b"no_such_knob", 1) != 0
    assert b"unknown key" in lib.cagc_last_error()


# ---------------------------------------------------------------------------------------------------
# register-direct weight gradient (csrc/conv_wgrad_rd.hip)
# ---------------------------------------------------------------------------------------------------
WG_CONFIGS = {"default": {}, "lds_kernels": {"wgrad_rd": 0}, "few_splits": {"wgrad_rd_wgs": 8}, "many_splits": {"wgrad_rd_wgs": 100000}}
WG_DEFAULTS = {"wgrad_rd": 1, "wgrad_rd_wgs": 0}


@pytest.fixture(params=list(WG_CONFIGS))
def wg_tuning(request):
    lib = _lib.load()
    for k, v in {**WG_DEFAULTS, **WG_CONFIGS[request.param]}.items():
        assert lib.cagc_set_tuning(k.encode(), v) == 0
    yield request.param
    for k, v in WG_DEFAULTS.items():
        lib.cagc_set_tuning(k.encode(), v)


# (B, cin, cout, H, W, up, modulated)
WGRAD_SHAPES = [(3, 20, 36, 16, 16, 0, True), (2, 77, 39, 32, 32, 0, True), (4, 154, 154, 16, 32, 0, True), (2, 128, 64, 64, 64, 0, False),
                (1, 512, 512, 32, 32, 0, False), (3, 20, 36, 16, 16, 1, True), (2, 154, 77, 32, 32, 1, True), (2, 77, 39, 64, 64, 1, True),
                (2, 128, 256, 32, 32, 1, False), (5, 39, 39, 48, 48, 0, True)]
# 1x1 convolutions (ResBlock skip, from-RGB; reference model.py:726-728, 758): (B, cin, cout, H, W, modulated)
WGRAD_1X1_SHAPES = [(2, 128, 256, 32, 32, False), (3, 3, 128, 64, 64, False), (2, 77, 39, 16, 48, True), (1, 512, 512, 16, 16, False),
                    (4, 20, 36, 32, 16, False)]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_weight_gradient_plain_and_transposed(shape, wg_tuning):
    """gW = scale * sum_b s[b,i] sum_p g[b,o,p(+k)] x[b,i,p(+k)]  (cagc_modconv_wgrad; plain and transposed-conv geometry, the
    latter from the phase-planar gradient) against autograd of the float64 convolution."""
    B, cin, cout, H, W, up, mod = shape
    torch.manual_seed(14)
    scale = 1.0 / math.sqrt(cin * 9)
    x = torch.randn(B, cin, H, W)
    s = (torch.rand(B, cin) + 0.5) if mod else None
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    xs = x.double() * (s.double()[:, :, None, None] if mod else 1.0)
    if up:
        gfull = torch.randn(B, cout, 2 * H + 1, 2 * W + 1)
        y = F.conv_transpose2d(xs, (w * scale).transpose(0, 1), stride=2)
        (gref,) = torch.autograd.grad(y, w, gfull.double())
        gdev = phase_planar(gfull, H, W).to(DEV)
    else:
        gz = torch.randn(B, cout, H, W)
        y = F.conv2d(xs, w * scale, padding=1)
        (gref,) = torch.autograd.grad(y, w, gz.double())
        gdev = gz.to(DEV)
    xg = x.to(DEV)
    sg = s.to(DEV) if mod else None
    n_ws = _lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, 3, up)
    ws = torch.empty(n_ws, device=DEV)
    gw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
    _lib.call("cagc_modconv_wgrad", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(gdev), _lib.ptr(xg), _lib.ptr(sg), B, cin, cout, H, W, 3, up,
              float(scale))
    assert rel(gw, gref) <= 2e-5, ("wgrad", wg_tuning, shape, rel(gw, gref))     # sums over B*H*W pixels of random products


@pytest.mark.parametrize("shape", WGRAD_1X1_SHAPES)
def test_weight_gradient_1x1(shape, wg_tuning):
    B, cin, cout, H, W, mod = shape
    torch.manual_seed(15)
    scale = 1.0 / math.sqrt(cin)
    x, gz = torch.randn(B, cin, H, W), torch.randn(B, cout, H, W)
    s = (torch.rand(B, cin) + 0.5) if mod else None
    xs = x.double() * (s.double()[:, :, None, None] if mod else 1.0)
    gref = scale * torch.einsum("bohw,bihw->oi", gz.double(), xs)[:, :, None, None]
    xg, gg = x.to(DEV), gz.to(DEV)
    sg = s.to(DEV) if mod else None
    ws = torch.empty(_lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, 1, 0), device=DEV)
    gw = torch.full((cout, cin, 1, 1), float("nan"), device=DEV)
    _lib.call("cagc_modconv_wgrad", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(gg), _lib.ptr(xg), _lib.ptr(sg), B, cin, cout, H, W, 1, 0, float(scale))
    assert rel(gw, gref) <= 2e-5, ("wgrad 1x1", wg_tuning, shape, rel(gw, gref))
