"""upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))   — API of reference op/upfirdn2d.py:145-156.

GPU tensors -> cagc_upfirdn2d (csrc/upfirdn2d.hip) with first- and second-order autograd (backward = the
same op with the flipped kernel and up<->down swapped, reference op/upfirdn2d.py:29-43,111-116); CPU tensors
-> composed PyTorch (zero-insert, pad/crop, conv with the flipped kernel, decimate)."""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from .. import _lib


def _out_size(n, up, p0, p1, k, down):
    return (n * up + p0 + p1 - k) // down + 1


def _launch(x4, kernel, up, down, pad, out_hw, out=None):
    """x4 [N,C,H,W] contiguous -> [N,C,oh,ow] (`out`: a caller-allocated contiguous result, e.g. one allocated on another stream)."""
    n, c, h, w = x4.shape
    kh, kw = kernel.shape
    oh, ow = out_hw
    if out is None:
        out = torch.empty(n, c, oh, ow, dtype=x4.dtype, device=x4.device)
    assert tuple(out.shape) == (n, c, oh, ow) and out.dtype == x4.dtype and out.is_contiguous()
    if x4.dtype != torch.float32:     # fp16 / fp64: generic any-dtype kernel, FIR taps in the tensors' element type
        k = kernel.to(x4.dtype).contiguous()
        with _lib.on_device(x4):
            _lib.call("cagc_upfirdn2d_any", _lib.ptr_any(out), _lib.ptr_any(x4), _lib.ptr_any(k), _lib.DTYPE_CODE[x4.dtype], n * c,
                      h, w, oh, ow, kh, kw, up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3])
        return out
    with _lib.on_device(x4):
        _lib.call("cagc_upfirdn2d", _lib.ptr(out), _lib.ptr(x4), _lib.ptr(kernel), n * c, h, w, oh, ow, kh, kw, up[0],
                  up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3])
    return out


class _UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        go = grad_output.contiguous()
        gin = _launch(go, grad_kernel, down, up, g_pad, (in_size[2], in_size[3]))
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return gin

    @staticmethod
    def backward(ctx, gg_input):
        (kernel,) = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        gg_out = _launch(gg_input.contiguous(), kernel, up, down, pad, out_size)
        return gg_out, None, None, None, None, None, None, None, None


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        x = input.contiguous()
        kernel = kernel.contiguous()
        kh, kw = kernel.shape
        n, c, h, w = x.shape
        oh = _out_size(h, up[1], pad[2], pad[3], kh, down[1])
        ow = _out_size(w, up[0], pad[0], pad[1], kw, down[0])
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        # padding of the adjoint op (reference op/upfirdn2d.py:111-116)
        g_pad = (kw - pad[0] - 1, w * up[0] - ow * down[0] + pad[0] - up[0] + 1,
                 kh - pad[2] - 1, h * up[1] - oh * down[1] + pad[2] - up[1] + 1)
        ctx.cfg = (up, down, pad, g_pad, tuple(x.shape), (oh, ow))
        return _launch(x, kernel, up, down, pad, (oh, ow))

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        up, down, pad, g_pad, in_size, out_size = ctx.cfg
        gin = _UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size)
        return gin, None, None, None, None


def _upfirdn2d_cpu(x, kernel, up, down, pad):
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    planes = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = planes.new_zeros(n * c, 1, h, up, w, up)
        z[:, :, :, 0, :, 0] = planes
        planes = z.reshape(n * c, 1, h * up, w * up)
    planes = F.pad(planes, [pad[0], pad[1], pad[0], pad[1]])  # negative = crop
    y = F.conv2d(planes, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(x.dtype))
    if down > 1:
        y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if input.device.type == "cpu":
        return _upfirdn2d_cpu(input, kernel, up, down, pad)
    if input.dtype not in (torch.float32, torch.float16, torch.float64):
        raise RuntimeError(f"upfirdn2d (HIP): unsupported dtype {input.dtype}")
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
