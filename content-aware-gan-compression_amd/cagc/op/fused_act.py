"""fused bias + LeakyReLU(0.2) * sqrt(2)   — API of reference op/fused_act.py:87-119.

GPU tensors -> cagc_fused_bias_act_{fwd,bwd,bwd2} (csrc/elementwise.hip); first- and second-order autograd
(the R1 and path-length regularisers differentiate through this op twice).  CPU tensors ->
composed PyTorch, like the reference's CPU branch."""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

from .. import _lib


def _view3(t):
    """[N, C, *] -> (outer, C, inner)."""
    inner = 1
    for d in t.shape[2:]:
        inner *= d
    return t.shape[0], t.shape[1], inner


class _LReLUBackward(Function):
    """gx = gout * gate(out) * scale, gbias = sum(gx) fused into the same pass; differentiable once more."""

    @staticmethod
    def forward(ctx, grad_output, out, has_bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        go = grad_output.contiguous()
        gx = torch.empty_like(go)
        outer, C, inner = _view3(go)
        gbias = torch.zeros(C, dtype=go.dtype, device=go.device) if has_bias else None
        with _lib.on_device(go):
            _lib.call("cagc_fused_bias_act_bwd", _lib.ptr(gx), _lib.ptr(gbias), _lib.ptr(go), _lib.ptr(out), outer, C,
                      inner, negative_slope, scale)
        return gx, (gbias if has_bias else go.new_empty(0))

    @staticmethod
    def backward(ctx, gg_input, gg_bias):
        (out,) = ctx.saved_tensors
        ggi = gg_input.contiguous()
        ggb = gg_bias.contiguous() if (gg_bias is not None and gg_bias.numel() > 0) else None
        ggo = torch.empty_like(ggi)
        outer, C, inner = _view3(ggi)
        with _lib.on_device(ggi):
            _lib.call("cagc_fused_bias_act_bwd2", _lib.ptr(ggo), _lib.ptr(ggi), _lib.ptr(ggb), _lib.ptr(out), outer, C,
                      inner, ctx.negative_slope, ctx.scale)
        return ggo, None, None, None, None


class _LReLU(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        x = input.contiguous()
        out = torch.empty_like(x)
        outer, C, inner = _view3(x)
        b = bias.contiguous() if bias is not None else None
        if b is not None and b.numel() != C:
            raise RuntimeError(f"fused_leaky_relu: bias has {b.numel()} elements, input has {C} channels")
        with _lib.on_device(x):
            _lib.call("cagc_fused_bias_act_fwd", _lib.ptr(out), _lib.ptr(x), _lib.ptr(b), outer, C, inner, negative_slope,
                      scale)
        ctx.save_for_backward(out)
        ctx.has_bias = bias is not None
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        gx, gb = _LReLUBackward.apply(grad_output, out, ctx.has_bias, ctx.negative_slope, ctx.scale)
        return gx, (gb if ctx.has_bias else None), None, None


# ---------------------------------------------------------------------------------------------------
# fp16 / fp64: the generic any-dtype kernels (cagc_fused_bias_act_any) with the same first- and second-order autograd
# ---------------------------------------------------------------------------------------------------
def _any(mode, a, bias, ref, negative_slope, scale):
    a = a.contiguous()
    out = torch.empty_like(a)
    outer, C, inner = _view3(a)
    with _lib.on_device(a):
        _lib.call("cagc_fused_bias_act_any", _lib.ptr_any(out), _lib.ptr_any(a), _lib.ptr_any(bias), _lib.ptr_any(ref),
                  _lib.DTYPE_CODE[a.dtype], mode, outer, C, inner, float(negative_slope), float(scale))
    return out


class _LReLUBackwardAny(Function):
    @staticmethod
    def forward(ctx, grad_output, out, has_bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        gx = _any(1, grad_output, None, out, negative_slope, scale)
        dims = [0] + list(range(2, gx.ndim))
        return gx, (gx.sum(dims) if has_bias else gx.new_empty(0))      # grad_bias as the reference: a separate sum (:33-39)

    @staticmethod
    def backward(ctx, gg_input, gg_bias):
        (out,) = ctx.saved_tensors
        ggb = gg_bias.contiguous() if (gg_bias is not None and gg_bias.numel() > 0) else None
        return _any(2, gg_input, ggb, out, ctx.negative_slope, ctx.scale), None, None, None, None


class _LReLUAny(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        b = bias.contiguous().to(input.dtype) if bias is not None else None
        if b is not None and b.numel() != input.shape[1]:
            raise RuntimeError(f"fused_leaky_relu: bias has {b.numel()} elements, input has {input.shape[1]} channels")
        out = _any(0, input, b, None, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.has_bias = bias is not None
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        gx, gb = _LReLUBackwardAny.apply(grad_output, out, ctx.has_bias, ctx.negative_slope, ctx.scale)
        return gx, (gb if ctx.has_bias else None), None, None


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type == "cpu":
        if bias is not None:
            input = input + bias.reshape((1, -1) + (1,) * (input.ndim - 2))
        # the reference's CPU branch pins the slope to 0.2 whatever is passed (op/fused_act.py:110,116);
        # every caller passes 0.2, so honouring the argument is equivalent and matches its GPU branch
        return F.leaky_relu(input, negative_slope=negative_slope) * scale
    if input.dtype in (torch.float16, torch.float64):      # the reference dispatches fp32 / fp64 / fp16 (kernel.cu:79)
        return _LReLUAny.apply(input, bias, negative_slope, scale)
    if input.dtype != torch.float32:
        raise RuntimeError(f"fused_leaky_relu (HIP): unsupported dtype {input.dtype}")
    return _LReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
