"""Winograd-domain fused-phase stride-2 transposed convolution (csrc/conv_up25.hip: 25 instead of 36 position-GEMMs per 2x2 tile of
positions) against float64 and against the kernels it would replace, through the C ABI.

Same entry points and shapes as tests/test_conv_up4_gpu.py (`cagc_modconv_up_fwd`, reference model.py:259-270; `cagc_conv3x3s2_dgrad`, data
gradient of model.py:693-706).  `cagc_set_tuning("up25", 1)` + `"up25_min_ksteps", 0` send the small shapes there; `"up25_launches"` proves
the kernel took the launch.  The transforms have coefficients 0 / +-1 only: the float64 bar is the direct kernels' 5e-6 loosened to 2e-5
(one cancellation per product instead of none)."""
import math

import pytest
import torch
from torch.nn import functional as F

from cagc import _lib
from cagc.op import modconv as mc
from test_conv_up4_gpu import UP_SHAPES, S2_SHAPES, phase_planar, rel, no_spin_timeout

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5


def took(before):
    assert _lib.get_tuning("up25_launches") == before + 1, "the launch did not reach conv_up25.hip"


@pytest.mark.parametrize("modulated", [True, False])
@pytest.mark.parametrize("shape", UP_SHAPES)
def test_up_fwd_winograd_phase_kernel(shape, modulated):
    B, cin, cout, H, W, lmin = shape
    torch.manual_seed(31)
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
    t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
    n0 = _lib.get_tuning("up25_launches")
    with _lib.tuning(up25=1, up25_min_ksteps=0, up25_lmin=lmin):
        _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(xg), _lib.ptr(wp_fwd), _lib.ptr(sg), B, cin, cout, H, W)
    torch.cuda.synchronize()
    took(n0)
    assert rel(t[..., :W + 1], tref[..., :W + 1]) <= TOL, (shape, modulated, rel(t[..., :W + 1], tref[..., :W + 1]))
    assert torch.isfinite(t).all()      # pad columns are written too (tiny values, not NaN)
    no_spin_timeout()


@pytest.mark.parametrize("shape", S2_SHAPES)
def test_stride2_data_gradient_winograd_phase_kernel(shape):
    B, cin, cout, H, lmin = shape
    torch.manual_seed(32)
    hb = H + 1
    pitch = (hb + 3) // 4 * 4
    ho = (hb - 3) // 2 + 1
    w = torch.randn(cout, cin, 3, 3)
    scale = 1.0 / math.sqrt(cin * 9)
    _, wp_bwd = mc.pack_plain_weights(w.to(DEV), scale, True)
    g = torch.randn(B, cout, ho, ho)
    gg = g.to(DEV)
    gref = F.conv_transpose2d(g.double(), w.double() * scale, stride=2)
    gx = torch.full((B, cin, hb, pitch), float("nan"), device=DEV)
    n0 = _lib.get_tuning("up25_launches")
    with _lib.tuning(up25=1, up25_min_ksteps=0, up25_lmin=lmin):
        _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gx), _lib.ptr(gg), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch)
    torch.cuda.synchronize()
    took(n0)
    assert rel(gx[..., :hb], gref) <= TOL, (shape, rel(gx[..., :hb], gref))
    no_spin_timeout()


def test_stream_k_handoff_is_bit_reproducible():
    torch.manual_seed(33)
    for (B, cin, cout, H) in [(2, 256, 128, 32), (16, 128, 256, 40)]:
        wt = torch.randn(1, cout, cin, 3, 3)
        wp_fwd, _, _ = mc.pack_weights(wt.to(DEV), True)
        x, s = torch.randn(B, cin, H, H, device=DEV), torch.rand(B, cin, device=DEV) + 0.5
        P = _lib.query("cagc_phase_pitch", H)
        first = None
        with _lib.tuning(up25=1, up25_min_ksteps=0, up25_lmin=2):
            for it in range(20):
                t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
                _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H)
                v = t[..., :H + 1].clone()
                if first is None:
                    first = v
                else:
                    assert torch.equal(v, first), (B, cin, cout, H, it)
        torch.cuda.synchronize()
    no_spin_timeout()


def test_output_beyond_two_gigabytes_is_cut_into_batch_chunks():
    """batch 64 at 256^2 (the saliency sweep's largest up layer): the phase-planar output is 2.2 GB, beyond the kernel's 32-bit buffer
    offsets — the launch runs as two batch chunks; every image must equal the same image computed in a small batch"""
    torch.manual_seed(34)
    B, cin, cout, H = 64, 16, 128, 128
    wt = torch.randn(1, cout, cin, 3, 3, device=DEV)
    wp_fwd, _, _ = mc.pack_weights(wt, True)
    x, s = torch.randn(B, cin, H, H, device=DEV), torch.rand(B, cin, device=DEV) + 0.5
    P = _lib.query("cagc_phase_pitch", H)
    t = torch.full((B, cout, 4, H + 1, P), float("nan"), device=DEV)
    assert t.numel() * 4 > 2 ** 31
    n0 = _lib.get_tuning("up25_launches")
    with _lib.tuning(up25=1, up25_min_ksteps=0):
        _lib.call("cagc_modconv_up_fwd", _lib.ptr(t), _lib.ptr(x), _lib.ptr(wp_fwd), _lib.ptr(s), B, cin, cout, H, H)
        assert _lib.get_tuning("up25_launches") == n0 + 2
        for b0 in (0, 31, 32, 63):
            tb = torch.full((1, cout, 4, H + 1, P), float("nan"), device=DEV)
            xb, sb = x[b0:b0 + 1].contiguous(), s[b0:b0 + 1].contiguous()
            _lib.call("cagc_modconv_up_fwd", _lib.ptr(tb), _lib.ptr(xb), _lib.ptr(wp_fwd), _lib.ptr(sb), 1, cin, cout, H, H)
            assert rel(t[b0:b0 + 1, ..., :H + 1], tb[..., :H + 1]) <= 1e-6, b0
    torch.cuda.synchronize()
    no_spin_timeout()



def test_streamk_error_word_is_polled_by_the_training_step():
    """a bounded stream-K spin that gives up used to leave a sticky DEVICE word nobody read on the step's path.  The word now
    lives in host-mapped memory (per device); KDStep.g_step / GraphedKDStep.replay read it every step without synchronising and raise.
    (tests/test_bench_multirank_gpu.py is where it fired for real: two processes' persistent kernels on one GPU.)"""
    from cagc import kd
    from cagc.op import modconv as mc
    assert _lib.get_tuning("streamk_error_nosync") == 0 and _lib.get_tuning("up4_error") == 0
    mc.check_streamk_error("cuda")
    student, teacher, disc = kd.build_synthetic_workload(256, "cuda", seed=1)
    step = kd.KDStep(student, teacher, disc)
    mask = kd.ellipse_mask(2, 256, "cuda")
    step.sample_and_step(2, mask)
    try:
        _lib.set_tuning("streamk_error_test", 1)
        assert _lib.get_tuning("streamk_error_nosync") == 1
        with pytest.raises(RuntimeError, match="stream-K"):
            mc.check_streamk_error("cuda", sync=True)
        with pytest.raises(RuntimeError, match="stream-K"):
            step.sample_and_step(2, mask)
    finally:
        _lib.set_tuning("streamk_error_test", 0)
    step.sample_and_step(2, mask)
