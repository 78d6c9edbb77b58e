"""Plain (un-fused) convolution on libcagc as a CLOSED family of autograd Functions, differentiable to any order.

The first-order hot path uses fused ops whose backward is final (`once_differentiable`).  The second-order passes of the
training iteration — R1 (reference train.py:194-200, 264-278: gradient penalty through the discriminator) and the
path-length regulariser (model.py:661-666, train.py:310-338) — differentiate a backward pass again.  A convolution is
bilinear in (input, weight), so its three maps

    F(x, w) = conv(x, scale*w)          D(g, w) = data gradient          W(g, x) = weight gradient

close under differentiation:   dF = (D, W)    dD = (F wrt g, W wrt w)    dW = (F wrt g, D wrt x).
Each is one launch of the MFMA kernels behind the C ABI (cagc_modconv_fwd / _dgrad / _wgrad, cagc_conv3x3s2_fwd /
_dgrad), so nothing on the second-order paths reaches MIOpen (the reference uses cuDNN's double-backward there).

modes:  "s1"  stride 1, 'same' padding, k = 1 or 3                                   (model.py:120, 282)
        "s2"  stride 2, no padding, k = 3, odd input size 2*Ho+1                    (model.py:276, 683-706)
              — and the transposed stride-2 conv of the up-sampling layers (model.py:259-267) is D("s2")
"""
import torch
from torch.autograd import Function

from .. import _lib


def _pack(w, scale, bwd):
    """[Cout,Cin,k,k] -> the MFMA A-operand packing (cagc_modconv_prep); bwd: the transposed/flipped one for D."""
    cout, cin, k, _ = w.shape
    w = w.detach().contiguous()
    n = _lib.query("cagc_modconv_packed_elems", cout if bwd else cin, cin if bwd else cout, k)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    with _lib.on_device(w):
        _lib.call("cagc_modconv_prep", None if bwd else _lib.ptr(wp), _lib.ptr(wp) if bwd else None, None, _lib.ptr(w),
                  cout, cin, k, float(scale))
    return wp


def _out_hw(mode, H, W):
    if mode == "s1":
        return H, W
    assert H % 2 == 1 and W % 2 == 1 and H >= 3 and W >= 3, "stride-2 closure conv wants an odd input size (2*Ho+1)"
    return (H - 3) // 2 + 1, (W - 3) // 2 + 1


def _fwd(x, w, mode, scale):
    x = x.contiguous()
    B, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    ho, wo = _out_hw(mode, H, W)
    out = torch.empty(B, cout, ho, wo, dtype=x.dtype, device=x.device)
    wp = _pack(w, scale, False)
    with _lib.on_device(x):
        if mode == "s1":
            _lib.call("cagc_modconv_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp), None, B, cin, cout, H, W, k, 0, None, None,
                      0, None, None, 0.2, 1.0)
        else:
            _lib.call("cagc_conv3x3s2_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(wp), B, cin, cout, H, W, W)
    return out


def _dgrad(g, w, mode, scale, in_hw):
    g = g.contiguous()
    B, cout, ho, wo = g.shape
    _, cin, k, _ = w.shape
    H, W = in_hw
    gx = torch.empty(B, cin, H, W, dtype=g.dtype, device=g.device)
    wp = _pack(w, scale, True)
    with _lib.on_device(g):
        if mode == "s1":
            _lib.call("cagc_modconv_dgrad", _lib.ptr(gx), None, _lib.ptr(g), _lib.ptr(wp), None, None, B, cin, cout, H, W, k)
        else:
            _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gx), _lib.ptr(g), _lib.ptr(wp), B, cin, cout, H, W, W)
    return gx


def wgrad_s1(g, x, k, scale):
    """[Cout,Cin,k,k] = scale * sum_{b,p} g[b,o,p] x[b,i,p(+tap)] — the un-modulated cagc_modconv_wgrad."""
    g, x = g.contiguous(), x.contiguous()
    B, cout, H, W = g.shape
    cin = x.shape[1]
    gw = torch.empty(cout, cin, k, k, dtype=g.dtype, device=g.device)
    ws = torch.empty(_lib.query("cagc_modconv_wgrad_workspace", B, cin, cout, H, W, k, 0), dtype=g.dtype, device=g.device)
    with _lib.on_device(g):
        _lib.call("cagc_modconv_wgrad", _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(g), _lib.ptr(x), None, B, cin, cout, H, W, k, 0,
                  float(scale))
    return gw


def wgrad_s2(g, xb, scale, in_pitch=None):
    """Weight gradient of the stride-2 3x3 conv: gw[o,i,ky,kx] = scale * sum g[b,o,y,x] xb[b,i,2y+ky,2x+kx].
    Same contraction as the transposed conv's weight gradient with the operand roles swapped: the (2Ho+1)-sized operand
    goes phase-planar (cagc_to_phase_planar) and takes the place of the transposed conv's output gradient."""
    g = g.contiguous()
    B, cout, ho, wo = g.shape
    cin, hb = xb.shape[1], xb.shape[2]
    pitch = xb.shape[3] if in_pitch is None else in_pitch
    assert hb == 2 * ho + 1
    P = _lib.query("cagc_phase_pitch", wo)
    t = torch.empty(B, cin, 4, ho + 1, P, dtype=g.dtype, device=g.device)
    gw_t = torch.empty(cin, cout, 3, 3, dtype=g.dtype, device=g.device)
    ws = torch.empty(_lib.query("cagc_modconv_wgrad_workspace", B, cout, cin, ho, wo, 3, 1), dtype=g.dtype, device=g.device)
    with _lib.on_device(g):
        _lib.call("cagc_to_phase_planar", _lib.ptr(t), _lib.ptr(xb), B * cin, ho, wo, pitch)
        _lib.call("cagc_modconv_wgrad", _lib.ptr(gw_t), _lib.ptr(ws), _lib.ptr(t), _lib.ptr(g), None, B, cout, cin, ho, wo, 3, 1,
                  float(scale))
    return gw_t.permute(1, 0, 2, 3).contiguous()


def _wgrad(g, x, mode, scale, k):
    if mode == "s1":
        return wgrad_s1(g, x, k, scale)
    return wgrad_s2(g, x.contiguous(), scale)


class ConvF(Function):
    """y = conv(x, scale * w)."""

    @staticmethod
    def forward(ctx, x, w, mode, scale):
        ctx.mode, ctx.scale = mode, scale
        ctx.save_for_backward(x, w)
        return _fwd(x, w, mode, scale)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = ConvD.apply(g, w, ctx.mode, ctx.scale, (x.shape[2], x.shape[3])) if ctx.needs_input_grad[0] else None
        gw = ConvW.apply(g, x, ctx.mode, ctx.scale, w.shape[-1]) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


class ConvD(Function):
    """gx = d <g, conv(x, scale*w)> / dx   (for mode "s2" this IS the stride-2 transposed convolution of g)."""

    @staticmethod
    def forward(ctx, g, w, mode, scale, in_hw):
        ctx.mode, ctx.scale = mode, scale
        ctx.save_for_backward(g, w)
        return _dgrad(g, w, mode, scale, in_hw)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        gg = ConvF.apply(ggx, w, ctx.mode, ctx.scale) if ctx.needs_input_grad[0] else None
        gw = ConvW.apply(g, ggx, ctx.mode, ctx.scale, w.shape[-1]) if ctx.needs_input_grad[1] else None
        return gg, gw, None, None, None


class ConvW(Function):
    """gw = d <g, conv(x, scale*w)> / dw."""

    @staticmethod
    def forward(ctx, g, x, mode, scale, k):
        ctx.mode, ctx.scale, ctx.k = mode, scale, k
        ctx.save_for_backward(g, x)
        return _wgrad(g, x, mode, scale, k)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        gg = ConvF.apply(x, ggw, ctx.mode, ctx.scale) if ctx.needs_input_grad[0] else None
        gx = ConvD.apply(g, ggw, ctx.mode, ctx.scale, (x.shape[2], x.shape[3])) if ctx.needs_input_grad[1] else None
        return gg, gx, None, None, None


def supported(x, weight, stride, padding):
    """Can conv2d(x, weight, stride, padding) run on the closure family?"""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4):
        return False
    k = weight.shape[-1]
    if weight.shape[-2] != k:
        return False
    if stride == 1:
        return k in (1, 3) and padding == k // 2
    if stride == 2 and padding == 0:
        if k == 3:
            return x.shape[2] % 2 == 1 and x.shape[3] % 2 == 1 and x.shape[2] >= 3 and x.shape[3] >= 3
        return k == 1
    return False


def conv2d(x, weight, scale=1.0, stride=1, padding=0):
    """F.conv2d(x, weight * scale, stride=stride, padding=padding) for the patterns `supported` accepts."""
    k = weight.shape[-1]
    if stride == 2 and k == 1:          # 1x1 stride 2 == 1x1 on the decimated input
        return ConvF.apply(x[:, :, ::2, ::2].contiguous(), weight, "s1", scale)
    return ConvF.apply(x, weight, "s1" if stride == 1 else "s2", scale)


def conv_transpose2d_s2(x, weight_t, scale=1.0):
    """F.conv_transpose2d(x, weight_t * scale, stride=2, padding=0) for a 3x3 `weight_t` [Cin,Cout,3,3]  ->  [B,Cout,2H+1,2W+1]."""
    H, W = x.shape[2], x.shape[3]
    return ConvD.apply(x, weight_t, "s2", scale, (2 * H + 1, 2 * W + 1))
