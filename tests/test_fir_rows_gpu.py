"""cagc_fir4x4_pitched through the C ABI against a float64 correlation — the row-streaming kernel (csrc/upfirdn2d.hip k_fir4_rows, taken for
outputs >= 32 x 32) and the tiled kernel it falls back to, on the cases the streaming form has to get right: every pad_x0 in [0, 4], a
NON-separable 4x4 kernel, source rows whose pitch padding holds NaNs (must not be read as data), widths that are not multiples of 4,
strips whose last group of 4 rows is partial, waves that straddle rows / strips / planes (65 column groups per row).
Replaces upfirdn2d_op.upfirdn2d(up = down = 1) in front of / behind the stride-2 conv (reference model.py:86-96, op/upfirdn2d.py:106-150)."""
import pytest
import torch
from torch.nn import functional as F

from cagc import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (planes, in_h, in_w, in_pitch, out_h, out_w, out_pitch, pad_x0, pad_y0)
CASES = [
    (6, 256, 256, 256, 257, 257, 260, 2, 2),     # D's blur in front of the stride-2 conv
    (6, 257, 257, 260, 256, 256, 256, 1, 1),     # its adjoint: pitched source, ragged width
    (5, 61, 45, 48, 64, 50, 52, 0, 3),
    (5, 61, 45, 48, 59, 44, 44, 1, 0),
    (7, 40, 66, 68, 41, 65, 68, 3, 1),
    (3, 37, 33, 36, 38, 36, 36, 4, 2),
    (130, 32, 32, 32, 33, 33, 36, 2, 2),         # many planes per workgroup
    (4, 16, 16, 16, 17, 17, 20, 2, 2),           # below the streaming kernel's size: tiled kernel
]


@pytest.mark.parametrize("case", CASES)
def test_fir4x4_pitched_vs_float64(case):
    planes, ih, iw, ip, oh, ow, op, px, py = case
    torch.manual_seed(5)
    k = torch.randn(4, 4)                                    # general, not an outer product
    x = torch.full((planes, ih, ip), float("nan"))
    x[:, :, :iw] = torch.randn(planes, ih, iw)
    # out[y, x] = sum_{i,j} kflip[i, j] * in[y - py + i, x - px + j]   (zeros outside the image)
    xp = F.pad(x[:, :, :iw].double().unsqueeze(1), (px, 8 + ow, py, 8 + oh))
    ref = F.conv2d(xp, torch.flip(k, (0, 1)).double().view(1, 1, 4, 4))[:, 0, :oh, :ow]
    out = torch.full((planes, oh, op), float("nan"), device=DEV)
    xd, kd = x.to(DEV), k.to(DEV)
    _lib.call("cagc_fir4x4_pitched", _lib.ptr(out), _lib.ptr(xd), _lib.ptr(kd), planes, ih, iw, ip, oh, ow, op, px, py)
    got = out.cpu()
    err = float((got[:, :, :ow].double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-6, (case, err)
    if op > ow:
        assert float(got[:, :, ow:].abs().max()) == 0.0, "pitch padding of the output is written as zero"


# 4x4 FIR + 2x decimation, pad (1,1) (the ResBlock skip's decimating blur, model.py:726-728 evaluated where the stride-2 1x1 conv samples;
# and the adjoint of `Upsample`, op/upfirdn2d.py:111-116) through cagc_upfirdn2d: the row-streaming kernel k_fir4_down2_rows (outputs
# >= 16 x 32 with widths % 4 == 0) and the tiled kernels it falls back to.  (planes, in_h, in_w)
DOWN2_CASES = [(6, 256, 256), (3, 128, 64), (130, 32, 64), (5, 34, 72), (4, 64, 40), (3, 16, 16), (2, 70, 66), (1, 8, 64)]


@pytest.mark.parametrize("case", DOWN2_CASES)
def test_fir4_down2_vs_float64_streaming_and_tiled(case):
    planes, ih, iw = case
    torch.manual_seed(6)
    k = torch.randn(4, 4)
    x = torch.randn(planes, ih, iw)
    oh, ow = (ih + 2 - 4) // 2 + 1, (iw + 2 - 4) // 2 + 1
    # upfirdn2d(up 1, down 2, pad (1,1)): out[oy,ox] = sum_{i,j} flip(k)[i,j] in[2oy-1+i, 2ox-1+j]
    ref = F.conv2d(F.pad(x.double().unsqueeze(1), (1, 1, 1, 1)), torch.flip(k, (0, 1)).double().view(1, 1, 4, 4), stride=2)[:, 0]
    assert tuple(ref.shape) == (planes, oh, ow)
    xd, kd = x.to(DEV), k.to(DEV)
    outs = []
    for rows in (1, 0):      # the default (streaming where eligible) and the tiled kernel: CAGC_FIR_ROWS is read once per process, so the
        out = torch.full((planes, oh, ow), float("nan"), device=DEV)      # tiled form is reached through a width the streaming form declines
        if rows == 0 and ow % 4 == 0 and oh >= 16 and ow >= 32:
            continue
        _lib.call("cagc_upfirdn2d", _lib.ptr(out), _lib.ptr(xd), _lib.ptr(kd), planes, ih, iw, oh, ow, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1)
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6, (case, rows, err)
        outs.append(out)
