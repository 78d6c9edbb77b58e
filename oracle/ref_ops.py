"""Oracle restatement of the reference operator layer (L1) and modulated conv — torch fp32, CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fir_kernel(taps, gain=1.0):
    """Normalised 2-D FIR from 1-D taps (outer product / sum) — reference model.py:27-35."""
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum() * gain


def fused_leaky_relu_ref(x, bias=None, negative_slope=0.2, scale=SQRT2):
    """y = leaky_relu(x + bias[c], 0.2) * scale — reference op/fused_act.py:104-119 (CPU branch;
    note it hard-codes slope 0.2; callers only ever pass 0.2) and the CUDA
    kernel's act=3 forward, op/fused_bias_act_kernel.cu:18-49."""
    if bias is not None:
        x = x + bias.reshape((1, -1) + (1,) * (x.ndim - 2))
    gate = x > 0
    if _GATES is not None:
        gate = _GATES.next(x, gate)
    return torch.where(gate, x, x * negative_slope) * scale


class gates:
    """Test hook around the oracle's LeakyReLU gates (in call order).  `with gates() as rec:` records, per activation,
    the pre-activation tensor and its gate; `with gates(force=[bool tensors])` evaluates the SAME network with the given
    gate pattern instead of its own.  Purpose: a random-init net has pre-activations at rounding distance from 0, where
    fp32 and float64 legitimately pick different sides; the parity tests prove every such disagreement is at rounding
    level and then compare all gradients on the common gate pattern, where they must agree to ~1e-6."""

    def __init__(self, force=None):
        self.force = None if force is None else list(force)
        self.pre, self.own, self.i = [], [], 0

    def next(self, x, own_gate):
        self.pre.append(x.detach())
        self.own.append(own_gate)
        g = own_gate
        if self.force is not None:
            g = self.force[self.i].to(own_gate.device).reshape(own_gate.shape)
        self.i += 1
        return g

    def __enter__(self):
        global _GATES
        assert _GATES is None
        _GATES = self
        return self

    def __exit__(self, *a):
        global _GATES
        _GATES = None


_GATES = None


def gate_disagreements(rec, gpu_gates, rounding=1e-5, max_fraction=1e-5):
    """Compare the oracle's own gates (a `gates()` recording, float64) with the HIP path's (sign of its activations, same
    order).  Asserts each disagreement sits at a pre-activation below `rounding` x the layer's scale and that they are
    few; returns their count."""
    n_dis, n_all = 0, 0
    assert len(rec.own) == len(gpu_gates), (len(rec.own), len(gpu_gates))
    for pre, own, gg in zip(rec.pre, rec.own, gpu_gates):
        gg = gg.reshape(own.shape).to(own.device)
        dis = own != gg
        n_all += dis.numel()
        k = int(dis.sum())
        if k:
            worst = float(pre[dis].abs().max()) / float(pre.abs().max())
            assert worst < rounding, f"LeakyReLU gate disagreement at |pre-activation| = {worst:.2e} of the layer scale"
            n_dis += k
    assert n_dis <= max(4, max_fraction * n_all), f"{n_dis} of {n_all} gates disagree"
    return n_dis


def upfirdn2d_ref(x, kernel, up=1, down=1, pad=(0, 0)):
    """Zero-insert upsample by `up`, pad/crop by (pad0, pad1) on both axes, correlate with the
    *flipped* FIR, keep every `down`-th sample — reference op/upfirdn2d.py:145-200.
    x: [N,C,H,W]; kernel [kh,kw]."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    planes = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = planes.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = planes
        planes = z
    # F.pad accepts negative amounts = crop (op/upfirdn2d.py:172-180)
    planes = F.pad(planes, [p0, p1, p0, p1])
    y = F.conv2d(planes, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(x.dtype))
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def equal_linear_ref(x, weight, bias, lr_mul=1.0, activation=False):
    """reference model.py:137-166: F.linear(x, W*scale) then (+ bias*lr_mul | fused lrelu)."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    y = F.linear(x, weight * scale)
    if activation:
        return fused_leaky_relu_ref(y, None if bias is None else bias * lr_mul)
    return y if bias is None else y + bias * lr_mul


def pixel_norm_ref(x):
    """reference model.py:23-24."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def modulated_conv2d_ref(x, style_w, weight, mod_weight, mod_bias, demodulate=True, upsample=False,
                         downsample=False, blur_taps=(1, 3, 3, 1)):
    """reference model.py:241-289 — per-sample modulated (and demodulated) grouped conv.

    x [B,Cin,H,W]; style_w [B,style_dim]; weight [1,Cout,Cin,k,k]; returns (out, s) with
    s the modulation scalars [B,1,Cin,1,1] (`return_style_scalars`)."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear_ref(style_w, mod_weight, mod_bias).reshape(b, 1, cin, 1, 1)       # :248
    wm = (1.0 / math.sqrt(cin * k * k)) * weight * s                                     # :249
    if demodulate:
        d = torch.rsqrt(wm.pow(2).sum([2, 3, 4]) + 1e-8)                                 # :252
        wm = wm * d.reshape(b, cout, 1, 1, 1)
    if upsample:                                                                         # :259-270
        wt = wm.transpose(1, 2).reshape(b * cin, cout, k, k)
        y = F.conv_transpose2d(x.reshape(1, b * cin, h, w), wt, stride=2, padding=0, groups=b)
        y = y.reshape(b, cout, y.shape[2], y.shape[3])
        p = (len(blur_taps) - 2) - (k - 1)
        y = upfirdn2d_ref(y, fir_kernel(blur_taps, 4.0), pad=((p + 1) // 2 + 1, p // 2 + 1))
    elif downsample:                                                                     # :272-278
        p = (len(blur_taps) - 2) + (k - 1)
        xb = upfirdn2d_ref(x, fir_kernel(blur_taps), pad=((p + 1) // 2, p // 2))
        y = F.conv2d(xb.reshape(1, b * cin, xb.shape[2], xb.shape[3]), wm.reshape(b * cout, cin, k, k),
                     stride=2, groups=b)
        y = y.reshape(b, cout, y.shape[2], y.shape[3])
    else:                                                                                # :280-284
        y = F.conv2d(x.reshape(1, b * cin, h, w), wm.reshape(b * cout, cin, k, k), padding=k // 2, groups=b)
        y = y.reshape(b, cout, h, w)
    return y, s
