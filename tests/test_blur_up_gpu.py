"""cagc_blur_up_fwd through the C ABI against a float64 evaluation: the blur behind the transposed conv (reference model.py:259-270:
conv_transpose2d -> Blur(pad (1,1)) -> [demodulation, noise, bias, LeakyReLU]) reading the phase-planar intermediate
T[B,C,4,H+1,P] (T_full[Y,X] = T[2(Y&1)+(X&1)][Y>>1][X>>1]): the 64-wide and 32-wide LDS-tiled kernels and the scalar one, every
epilogue form; NaNs in the planes' pitch padding must never be read."""
import pytest
import torch
from torch.nn import functional as F

from cagc import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, C, H, W)
CASES = [(4, 154, 64, 64), (2, 77, 128, 128), (16, 39, 32, 32), (3, 40, 32, 32), (2, 24, 8, 8), (1, 600, 64, 32), (2, 160, 36, 68)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["styled_per_sample_noise", "styled_shared_noise", "styled_no_noise", "linear"])
def test_blur_up_fwd_vs_float64(case, mode):
    B, C, H, W = case
    torch.manual_seed(9)
    P = _lib.query("cagc_phase_pitch", W)
    t = torch.full((B, C, 4, H + 1, P), float("nan"))
    t[..., :W + 1] = torch.randn(B, C, 4, H + 1, W + 1)
    fir = torch.randn(4, 4)
    # T_full [B,C,2H+2,2W+2]; out[Y,X] = sum_{a,b} flip(fir)[a,b] T_full[Y-1+a, X-1+b], Y < 2H, X < 2W
    tf = torch.zeros(B, C, 2 * H + 2, 2 * W + 2, dtype=torch.float64)
    for ph in range(4):
        tf[:, :, (ph >> 1)::2, (ph & 1)::2] = t[:, :, ph, :, :W + 1].double()
    full = F.conv2d(F.pad(tf.reshape(B * C, 1, 2 * H + 2, 2 * W + 2), (1, 0, 1, 0)), torch.flip(fir, (0, 1)).double().view(1, 1, 4, 4))
    ref = full[:, 0, :2 * H, :2 * W].reshape(B, C, 2 * H, 2 * W)
    d = torch.rand(B, C) + 0.5
    bias, nw = 0.1 * torch.randn(C), torch.tensor([0.3])
    nb = B if mode == "styled_per_sample_noise" else 1
    noise = torch.randn(nb, 1, 2 * H, 2 * W)
    styled = mode != "linear"
    ref = ref * d.double()[:, :, None, None]
    if styled:
        pre = ref + bias.double()[None, :, None, None] + (0.3 * noise.double() if mode != "styled_no_noise" else 0.0)
        ref = F.leaky_relu(pre, 0.2) * 2 ** 0.5
    td, fd, dd, bd, nwd, nd = (x.to(DEV) for x in (t, fir, d, bias, nw, noise))
    out = torch.full((B, C, 2 * H, 2 * W), float("nan"), device=DEV)
    has_noise = styled and mode != "styled_no_noise"
    _lib.call("cagc_blur_up_fwd", _lib.ptr(out), _lib.ptr(td), _lib.ptr(fd), _lib.ptr(dd), _lib.ptr(nd) if has_noise else None, nb if has_noise else 0,
              _lib.ptr(nwd) if has_noise else None, _lib.ptr(bd) if styled else None, B, C, H, W, 0.2, 2 ** 0.5)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-6, (case, mode, err)


@pytest.mark.parametrize("case", CASES + [(8, 128, 128, 128)])
def test_blur_up_bwd_vs_float64(case):
    """cagc_blur_up_bwd: the adjoint of the blur behind the transposed conv, gz [B,C,2H,2W] -> phase-planar gT [B,C,4,H+1,P]
    (gT_full[Yt,Xt] = sum_{i,j} fir[i,j] gz[Yt-2+i, Xt-2+j], Yt <= 2H, Xt <= 2W; the planes' extra row / column and the pitch padding are
    written as zero): round 6's row-streaming kernel and the tiled kernels the smaller cases fall to."""
    B, C, H, W = case
    torch.manual_seed(10)
    P = _lib.query("cagc_phase_pitch", W)
    gz = torch.randn(B, C, 2 * H, 2 * W)
    fir = torch.randn(4, 4)
    full = F.conv2d(F.pad(gz.double().reshape(B * C, 1, 2 * H, 2 * W), (2, 2, 2, 2)), fir.double().view(1, 1, 4, 4))      # [.., 2H+1, 2W+1]
    tf = torch.zeros(B * C, 2 * H + 2, 2 * W + 2, dtype=torch.float64)
    tf[:, :2 * H + 1, :2 * W + 1] = full[:, 0]
    ref = torch.zeros(B * C, 4, H + 1, P, dtype=torch.float64)
    for ph in range(4):
        ref[:, ph, :, :W + 1] = tf[:, (ph >> 1)::2, (ph & 1)::2]
    gt = torch.full((B, C, 4, H + 1, P), float("nan"), device=DEV)
    gzd, fd = gz.to(DEV), fir.to(DEV)
    _lib.call("cagc_blur_up_bwd", _lib.ptr(gt), _lib.ptr(gzd), _lib.ptr(fd), B, C, H, W)
    got = gt.cpu().double().reshape(B * C, 4, H + 1, P)
    assert torch.isfinite(got).all(), "an entry of the phase planes (incl. padding) was not written"
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 2e-6, (case, err)
    assert float(got[:, :, :, W + 1:].abs().max()) == 0.0 and float(got[:, 2:, H, :].abs().max()) == 0.0 and float(got[:, 1::2, :, W].abs().max()) == 0.0
