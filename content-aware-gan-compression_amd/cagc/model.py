"""StyleGAN2 Generator / Discriminator with the reference's Python API surface and state-dict contract
(reference model.py), running on the hand-written gfx950 kernels for GPU tensors.

What is kept byte-for-byte: class names, constructor arguments and defaults, forward keyword arguments,
attribute names read by the reference's callers (`.size`, `.n_latent`, `.style`, `.input`, `.conv1`, `.convs`,
`.to_rgbs`, `.to_rgb1`, `.noises`, `module.conv.weight`, `.conv.modulation`, `.conv.scale`, `.conv.demodulate`,
`.conv.out_channel`), parameter / buffer names and their registration order (Util/mask_util.py and
Util/network_util.py parse `state_dict()` by substring and position).

What is different underneath: StyledConv = ONE fused op (modulated conv on the fp32 MFMA with the
modulation applied to the staged input tile, demodulation + noise + bias + LeakyReLU in the epilogue); the
upsampling conv writes a phase-planar intermediate consumed by a fused blur+epilogue kernel; ToRGB is one
streaming kernel including the skip upsample; nothing materialises per-sample weights [B,Cout,Cin,k,k].
"""
import math
import random

import torch
from torch import autograd, nn
from torch.nn import functional as F

from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
from .op import modconv as mc


class PixelNorm(nn.Module):
    """x * rsqrt(mean_c(x^2) + 1e-8) — reference model.py:14-24."""

    def forward(self, input):
        return mc.pixel_norm(input)


def make_kernel(k):
    """1-D taps -> normalised 2-D FIR — reference model.py:27-35."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def _resample_pad(ktaps, factor, extra=0):
    p = ktaps - factor + extra
    return p


class Upsample(nn.Module):
    """reference model.py:38-56."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """reference model.py:59-77."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    """reference model.py:80-96."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer("kernel", k)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """reference model.py:99-134.  Inside the discriminator's ConvLayer the convolution runs on the hand-written kernels
    (ConvLayer.forward dispatches on the layer pattern); this module's own forward is the composed formulation used for
    CPU tensors and for layer shapes no kernel covers."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        if input.is_cuda:
            from .op import conv_closure as cc
            if cc.supported(input, self.weight, self.stride, self.padding):
                out = cc.conv2d(input, self.weight, self.scale, self.stride, self.padding)
                return out if self.bias is None else out + self.bias.view(1, -1, 1, 1)
            mc.warn_stock_conv_once(f"EqualConv2d(k={self.weight.shape[-1]}, stride={self.stride}, padding={self.padding}, "
                                    f"dtype={input.dtype})")
        return F.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]}, {self.weight.shape[2]},"
                f" stride={self.stride}, padding={self.padding})")


class EqualLinear(nn.Module):
    """reference model.py:137-171."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _frozen_scaled(self):
        """(weight * scale, bias * lr_mul) of a frozen layer (teacher, D on the generator step), cached against the
        tensors' version counters — two elementwise launches per layer and step that never change."""
        w, b = self.weight, self.bias
        key = (w._version, w.data_ptr(), w.device, None if b is None else (b._version, b.data_ptr()))
        c = getattr(self, "_scaled", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                c = (key, (w * self.scale), None if b is None else (b * self.lr_mul))
            self._scaled = c
        return c[1], c[2]

    def _apply(self, fn, *a, **k):
        self._scaled = None
        return super()._apply(fn, *a, **k)

    def forward(self, input):
        w, b = self.weight, self.bias
        if mc.map_linear_ok(input, self):     # few rows: GEMM + bias (+ LeakyReLU) as one launch, the whole backward as another
            return mc._MapLinear.apply(input, w, b, self.scale, self.lr_mul, bool(self.activation))
        if input.is_cuda and not mc.composed_active():
            mc.warn_library_gemm_once(f"EqualLinear({w.shape[1]}, {w.shape[0]}) on a {tuple(input.shape)} {input.dtype} input")
        # the cache is only for FROZEN layers (requires_grad False: teacher, D on the generator step).  A layer that
        # merely runs under no_grad (g_ema sampling) is not cached: the reference's EMA updates weights through `.data`
        # (train.py:129), which does not bump the version counter the cache is validated against.
        frozen = not (w.requires_grad or (b is not None and b.requires_grad))
        if frozen and not torch.jit.is_tracing():
            ws, bs = self._frozen_scaled()
            if self.activation:
                return fused_leaky_relu(F.linear(input, ws), bs)
            return F.linear(input, ws, bias=bs)
        if input.dim() == 2 and b is not None:
            # scale folded into the GEMM (alpha) instead of a weight-sized elementwise pass; lr_mul == 1 needs no pass
            be = b if self.lr_mul == 1 else b * self.lr_mul
            if self.activation:
                return fused_leaky_relu(torch.addmm(be, input, w.t(), beta=0, alpha=self.scale), be)
            return torch.addmm(be, input, w.t(), alpha=self.scale)
        if self.activation:
            out = F.linear(input, self.weight * self.scale)
            return fused_leaky_relu(out, self.bias * self.lr_mul)
        return F.linear(input, self.weight * self.scale, bias=self.bias * self.lr_mul)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


def invalidate_caches(module):
    """Drop every cached packed / scaled weight below `module`.  The caches of FROZEN layers are validated against the
    weight tensor's version counter and storage; an update that bypasses the counter (`param.data.mul_()`, a raw-pointer
    write, a HIP-graph replay that rewrites a frozen weight) must be followed by this call.  `cagc.kd.accumulate` and
    `load_checkpoint` call it; `load_state_dict` / optimiser steps bump the counter themselves."""
    for m in module.modules():
        for attr in ("_scaled", "_packed", "_wino", "_packed_wino", "_packed_gemm"):
            if getattr(m, attr, None) is not None:
                setattr(m, attr, None)


class ScaledLeakyReLU(nn.Module):
    """reference model.py:174-183."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ModulatedConv2d(nn.Module):
    """reference model.py:186-289.  forward(input, style, return_style_scalars=False)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:   # `blur` is registered before `weight`/`modulation`: state-dict order (App. C)
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._packed = None  # (weight version, data_ptr, wp_fwd, wp_bwd, wsq) for frozen weights
        self._wino = None

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    # -- HIP plumbing ----------------------------------------------------------------------------
    def _hip_pattern(self):
        """The layer pattern has a hand-written kernel (independent of the input tensor)."""
        return (not self.downsample and self.kernel_size in (1, 3)
                and not (self.upsample and (self.kernel_size != 3 or tuple(self.blur.kernel.shape) != (4, 4)
                                            or self.blur.pad != (1, 1))))

    def _hip_eligible(self, input):
        return mc.use_hip(input) and input.dtype == torch.float32 and self._hip_pattern()

    def packed_weights(self):
        """Packed GEMM operands of `weight` (csrc/conv_igemm.hip k_pack_weights).  Trainable weights are
        re-packed every forward (they change every optimiser step); frozen weights (teacher, g_ema, D-side
        use) are cached and re-validated against the tensor's version counter and storage."""
        w = self.weight
        need_bwd = torch.is_grad_enabled()
        if w.requires_grad:     # never cached, also under no_grad: `.data` updates (reference EMA) bypass `_version`
            return mc.pack_weights(w, need_bwd)
        key = (w._version, w.data_ptr(), w.device)
        if self._packed is None or self._packed[0] != key or (need_bwd and self._packed[1][1] is None):
            self._packed = (key, mc.pack_weights(w, need_bwd))
            self._wino = None
        return self._packed[1]

    def wino_weights(self, H, W):
        """Winograd-domain weights for the plain 3x3 forward (None when the layer / size is not eligible)."""
        if self.upsample or self.downsample or self.kernel_size != 3 or not mc.wino_ok(H, W):
            return None
        w = self.weight
        if w.requires_grad:
            return mc.pack_wino(w[0], self.scale, False)
        key = (w._version, w.data_ptr(), w.device)
        if getattr(self, "_wino", None) is None or self._wino[0] != key:
            self._wino = (key, mc.pack_wino(w[0], self.scale, False))
        return self._wino[1]

    def wino_weights_bwd(self, H, W):
        """Winograd-domain weights of the data gradient (flipped taps, swapped channel roles) — only for trainable
        weights under autograd, i.e. when a backward pass will run (None otherwise)."""
        w = self.weight
        if (self.upsample or self.downsample or self.kernel_size != 3 or not mc.wino_ok(H, W) or not mc.WINO_DGRAD
                or not (torch.is_grad_enabled() and w.requires_grad)):
            return None
        return mc.pack_wino(w[0], self.scale, True)

    def prepared(self, H, W):
        """(wp_fwd, wp_bwd, wsq, up_wino, up_wino_bwd) for an input of H x W.  A trainable weight changes every optimiser
        step: all five are rebuilt by one launch (mc.prep_all); frozen weights come from the caches above."""
        w = self.weight
        if w.requires_grad:
            need_bwd = torch.is_grad_enabled()
            wino = (not self.upsample and not self.downsample and self.kernel_size == 3 and mc.wino_ok(H, W))
            return mc.prep_all(w, need_bwd, wino, wino and need_bwd and mc.WINO_DGRAD)
        wp_fwd, wp_bwd, wsq = self.packed_weights()
        return wp_fwd, wp_bwd, wsq, self.wino_weights(H, W), self.wino_weights_bwd(H, W)

    def invalidate_packed(self):
        self._packed = None
        self._wino = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._wino = None
        return super()._apply(fn, *a, **k)

    def forward(self, input, style, return_style_scalars=False):
        batch, in_channel = input.shape[0], input.shape[1]
        s = self.modulation(style)                                        # [B, Cin]
        if self._hip_eligible(input):
            wp_fwd, wp_bwd, wsq, up_w, up_wb = self.prepared(input.shape[2], input.shape[3])
            out = mc._ModConv.apply(input, self.weight, s, wsq if self.demodulate else None, None, None, None, wp_fwd, wp_bwd,
                                    self.blur.kernel if self.upsample else None, False, self.upsample, up_w, up_wb)
        else:
            out = mc.modconv_composed(input, self.weight, s, self.demodulate, self.upsample, self.downsample,
                                      self.blur.kernel if (self.upsample or self.downsample) else None,
                                      self.blur.pad if (self.upsample or self.downsample) else None)
        if return_style_scalars:
            return out, s.view(batch, 1, in_channel, 1, 1)
        return out


class NoiseInjection(nn.Module):
    """reference model.py:292-303."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    """reference model.py:306-320."""

    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """reference model.py:323-367: ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU; one fused op on the GPU."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def _fused_pattern(self):
        return (self.conv._hip_pattern() and self.activate.bias is not None
                and self.activate.negative_slope == 0.2 and abs(self.activate.scale - 2 ** 0.5) < 1e-12)

    def forward(self, input, style, return_style_scalars=False, noise=None, _s=None, _prep=None):
        """`_s`: the layer's modulation vector when the caller already has it (Generator's modulation bank); `_prep`: its
        prepared operands (wp_fwd, wp_bwd, wsq, up_wino, up_wino_bwd, d) from the generator's one-launch banks."""
        conv = self.conv
        fused = conv._hip_eligible(input) and self._fused_pattern()
        if fused:
            batch, cin, h, w = input.shape
            s = _s if _s is not None else conv.modulation(style)
            d_pre = None
            if _prep is not None and _s is not None:
                wp_fwd, wp_bwd, wsq, up_w, up_wb, d_pre = _prep
            else:
                wp_fwd, wp_bwd, wsq, up_w, up_wb = conv.prepared(h, w)
            oh, ow = (2 * h, 2 * w) if conv.upsample else (h, w)
            if noise is None:
                noise = input.new_empty(batch, 1, oh, ow).normal_()
            elif noise.shape[0] not in (1, batch) or tuple(noise.shape[1:]) != (1, oh, ow):
                noise = noise.expand(batch, 1, oh, ow)
            out = mc._ModConv.apply(input, conv.weight, s, wsq if conv.demodulate else None, noise, self.noise.weight, self.activate.bias, wp_fwd,
                                    wp_bwd, conv.blur.kernel if conv.upsample else None, True, conv.upsample, up_w, up_wb,
                                    d_pre if conv.demodulate else None)
            styles = s.view(batch, 1, cin, 1, 1)
        else:
            if return_style_scalars:
                out, styles = conv(input, style, True)
            else:
                out, styles = conv(input, style), None
            out = self.activate(self.noise(out, noise=noise))
        if return_style_scalars:
            return out, styles
        return out


class ToRGB(nn.Module):
    """reference model.py:370-395."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, return_style_scalars=False, _s=None, _chain=(-1, False)):
        conv = self.conv
        up = getattr(self, "upsample", None)
        fused = (mc.use_hip(input) and input.dtype == torch.float32
                 and (skip is None or (up is not None and up.factor == 2 and tuple(up.kernel.shape) == (4, 4)
                                       and up.pad == (2, 1) and skip.shape[2] * 2 == input.shape[2])))
        if fused:
            s = _s if _s is not None else conv.modulation(style)
            out = mc._ToRGB.apply(input, conv.weight, s, self.bias, skip, up.kernel if skip is not None else None, _chain[0], _chain[1])
            styles = s.view(input.shape[0], 1, input.shape[1], 1, 1)
        else:
            if skip is not None:
                mc.torgb_join(skip, _chain[0])      # an earlier ToRGB of this chain may have produced `skip` on the side stream
            out, styles = conv(input, style, True)
            out = out + self.bias
            if skip is not None:
                out = out + self.upsample(skip)
        if return_style_scalars:
            return out, styles
        return out


class Generator(nn.Module):
    """reference model.py:398-666 — same constructor (incl. `generator_net_shape` for pruned students) and
    forward keywords."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 generator_net_shape=None):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp,
                                                              activation="fused_lrelu") for _ in range(n_mlp)])
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier, 512: 32 * channel_multiplier,
                         1024: 16 * channel_multiplier}
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        if generator_net_shape is None:   # channel count entering every styled conv, plus the last output
            res = [4] + [2 ** i for i in range(3, self.log_size + 1) for _ in range(2)]
            shape = [self.channels[4]] + [self.channels[r] for r in res]
        else:
            shape = list(generator_net_shape)
        self.input = ConstantInput(shape[0])
        self.conv1 = StyledConv(shape[0], shape[1], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(shape[1], style_dim, upsample=False)
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            r = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** r, 2 ** r))
        n_blocks = (len(shape) // 2 - 1) if generator_net_shape is not None else (self.log_size - 2)
        for i in range(1, n_blocks + 1):
            self.convs.append(StyledConv(shape[2 * i - 1], shape[2 * i], 3, style_dim, upsample=True,
                                         blur_kernel=blur_kernel))
            self.convs.append(StyledConv(shape[2 * i], shape[2 * i + 1], 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(shape[2 * i + 1], style_dim))
        self.n_latent = self.log_size * 2 - 2

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, noise_z, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                latent_styles=None, input_is_latent=False, noise=None, randomize_noise=True, PPL_regularize=False,
                return_rgb_list=False, return_style_scalars=False):
        if PPL_regularize:   # double backward: composed, twice-differentiable ops
            with mc.composed_autograd():
                image, latent = self._synthesize(noise_z, inject_index, truncation, truncation_latent, latent_styles,
                                                 input_is_latent, noise, randomize_noise, False, False, True)
                pl_noise = torch.randn_like(image) / math.sqrt(image.shape[2] * image.shape[3])
                grad, = autograd.grad(outputs=(image * pl_noise).sum(), inputs=latent, create_graph=True)
            path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
            return image, path_lengths
        return self._synthesize(noise_z, inject_index, truncation, truncation_latent, latent_styles, input_is_latent,
                                noise, randomize_noise, return_rgb_list, return_style_scalars, False)

    def _bank_styles(self, latent):
        """All modulation vectors s_l = modulation_l(latent[:, idx_l]) by one launch of the modulation bank (GPU fp32,
        first-order mode); None -> the layers evaluate their own EqualLinear."""
        if not (mc.use_hip(latent) and latent.dtype == torch.float32 and latent.dim() == 3):
            return None
        bank = getattr(self, "_mod_bank", None)
        # a shallow copy of the module (nn.DataParallel's replicate, copy.copy) inherits the bank through __dict__ while its
        # modulation layers are other objects, possibly on another device: such a bank is rebuilt for THIS module
        if bank and (bank.lins[0] is not self.conv1.conv.modulation or bank.lins[0].weight.device != latent.device):
            bank = None
        if bank is None:
            layers = [(self.conv1.conv.modulation, 0), (self.to_rgb1.conv.modulation, 1)]
            i = 1
            for blk, to_rgb in enumerate(self.to_rgbs):
                layers += [(self.convs[2 * blk].conv.modulation, i), (self.convs[2 * blk + 1].conv.modulation, i + 1),
                           (to_rgb.conv.modulation, i + 2)]
                i += 2
            bank = mc.ModulationBank(layers) if mc.ModulationBank.eligible(layers, self.style_dim) else False
            object.__setattr__(self, "_mod_bank", bank)      # not a submodule / not in the state dict
        if bank is False or latent.shape[1] <= max(bank.idx) or latent.shape[2] != 512:
            return None
        return bank(latent)

    def _styled_plan(self):
        """[(StyledConv, side of its input)] in forward order."""
        side = self.input.input.shape[2]
        plan = [(self.conv1, side)]
        for blk in range(len(self.to_rgbs)):
            plan.append((self.convs[2 * blk], side))
            side *= 2
            plan.append((self.convs[2 * blk + 1], side))
        return plan

    def _bank_prepare(self, latent, bank):
        """Everything the styled convs derive from their weights and modulation vectors, by TWO launches for the whole
        generator instead of two per layer (small per-GPU batches are launch-bound): the packed / Winograd-domain operands of
        every TRAINABLE layer (cagc_modconv_prep_bank; frozen layers keep their caches) and the demodulation factors
        d_l[b,o] of every layer (cagc_demod_bank).  Returns one (wp_fwd, wp_bwd, wsq, up_wino, up_wino_bwd, d) per styled conv,
        or None when a layer's pattern is outside the kernels' coverage (the layers then prepare themselves)."""
        from . import _lib
        plan = self._styled_plan()
        if bank is None or not all(m._fused_pattern() for m, _ in plan):
            return None
        B, dev = latent.shape[0], latent.device
        new = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        need_bwd = torch.is_grad_enabled()
        rows, pjobs, keep = [], [], []
        for m, side in plan:
            conv = m.conv
            w = conv.weight
            cout, cin, k = w.shape[1], w.shape[2], w.shape[-1]
            if w.requires_grad:
                wino = (not conv.upsample) and k == 3 and mc.wino_ok(side, side)
                wc = w.detach().contiguous()
                wp_fwd = new(_lib.query("cagc_modconv_packed_elems", cin, cout, k))
                wp_bwd = new(_lib.query("cagc_modconv_packed_elems", cout, cin, k)) if need_bwd else None
                wsq = torch.empty(cout, cin, dtype=torch.float32, device=dev)
                up_w = new(_lib.query("cagc_wino_packed_elems", cin, cout)) if wino else None
                up_wb = new(_lib.query("cagc_wino_packed_elems", cout, cin)) if (wino and need_bwd and mc.WINO_DGRAD) else None
                pjobs.append(_lib.PrepJob(_lib.ptr(wc), _lib.ptr(wp_fwd), _lib.ptr(wp_bwd), _lib.ptr(wsq), _lib.ptr(up_w), _lib.ptr(up_wb),
                                          cout, cin, k, float(conv.scale)))
                keep.append(wc)
            else:
                wp_fwd, wp_bwd, wsq = conv.packed_weights()
                up_w, up_wb = conv.wino_weights(side, side), conv.wino_weights_bwd(side, side)
            rows.append([wp_fwd, wp_bwd, wsq, up_w, up_wb, None])
        with _lib.on_device(latent):
            if pjobs:
                _lib.call("cagc_modconv_prep_bank", (_lib.PrepJob * len(pjobs))(*pjobs), len(pjobs))
            couts = [m.conv.weight.shape[1] for m, _ in plan]
            dflat = new(B * sum(couts))
            djobs, off = [], 0
            for i, ((m, _), row) in enumerate(zip(plan, rows)):
                if m.conv.demodulate:
                    s = bank[0 if i == 0 else 2 + 3 * ((i - 1) // 2) + (i - 1) % 2]
                    row[5] = dflat[off:off + B * couts[i]].view(B, couts[i])
                    djobs.append(_lib.DemodJob(_lib.ptr(row[5]), _lib.ptr(s), _lib.ptr(row[2]), m.conv.weight.shape[2], couts[i]))
                off += B * couts[i]
            if djobs:
                _lib.call("cagc_demod_bank", (_lib.DemodJob * len(djobs))(*djobs), len(djobs), B)
        return [tuple(r) for r in rows]

    def _synthesize(self, noise_z, inject_index, truncation, truncation_latent, latent_styles, input_is_latent, noise,
                    randomize_noise, return_rgb_list, return_style_scalars, want_latent):
        if input_is_latent:
            styles = latent_styles
        elif len(noise_z) > 1 and all(z.shape == noise_z[0].shape for z in noise_z):
            # one pass of the mapping network over the stacked latents (row-wise ops: identical values, half the launches)
            styles = list(self.style(torch.cat(list(noise_z), 0)).chunk(len(noise_z), 0))
        else:
            styles = [self.style(z) for z in noise_z]
        fresh_noise = noise is None and randomize_noise
        if noise is None:
            noise = ([None] * self.num_layers if randomize_noise
                     else [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)])
        if truncation < 1:
            styles = [truncation_latent + truncation * (w - truncation_latent) for w in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
        elif torch.is_tensor(inject_index):
            # device-side mixing index (static shapes: lets the whole step live in one HIP graph); value n_latent
            # reproduces the no-mixing case, values 1..n_latent-1 the reference's cat of the two repeated styles
            if mc.mix_latent_ok(styles[0], styles[1], inject_index):
                latent = mc._MixLatent.apply(styles[0], styles[1], inject_index, self.n_latent)
            else:
                pos = torch.arange(self.n_latent, device=styles[0].device).view(1, -1, 1)
                latent = torch.where(pos < inject_index.view(1, 1, 1), styles[0].unsqueeze(1), styles[1].unsqueeze(1))
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

        if fresh_noise and mc.use_hip(latent) and latent.dtype == torch.float32:
            # fresh N(0,1) noise for every layer (reference model.py:298-303: one `normal_()` per NoiseInjection) from ONE launch:
            # a single flat draw, viewed per layer as [B,1,H,W] — same distribution, 13 fewer launches per generator and step
            B = latent.shape[0]
            hw = [(2 ** ((i + 5) // 2)) ** 2 for i in range(self.num_layers)]
            flat = torch.empty(B * sum(hw), dtype=latent.dtype, device=latent.device).normal_()
            noise, off = [], 0
            for n in hw:
                side = int(round(n ** 0.5))
                noise.append(flat[off:off + B * n].view(B, 1, side, side))
                off += B * n
        rss = return_style_scalars
        styles_list = []
        bank = self._bank_styles(latent)            # every layer's modulation vector from ONE launch (or None)
        bs = (lambda k: bank[k]) if bank is not None else (lambda k: None)
        prep = self._bank_prepare(latent, bank)     # packed operands + demodulation factors of every layer: two launches (or None)
        pp = (lambda k: prep[k]) if prep is not None else (lambda k: None)
        out = self.input(latent)
        out = self.conv1(out, latent[:, 0], rss, noise=noise[0], _s=bs(0), _prep=pp(0))
        if rss:
            out, sc = out
            styles_list.append(sc)
        # The ToRGB chain (1x1 modulated conv + bias + up-sampled skip per resolution: reference model.py:380-395) only reads the
        # layer outputs: on the GPU its launches go to a side stream beside the styled convs (mc._ToRGB, mc.FORK_TORGB); the caller's
        # stream joins below.  `private`: the intermediate RGB outputs do not leave this function, so in backward each ToRGB's
        # skip gradient can stay on the side stream (only the next ToRGB's backward reads it).
        private = not return_rgb_list and not rss and not want_latent
        slot = -1
        fork_mode = mc.FORK_TORGB_MODE      # 1 every generator, 2 only under autograd (the student), 3 only without (teacher / EMA)
        if (mc.FORK_TORGB and out.is_cuda and not want_latent and out.shape[0] >= mc.FORK_TORGB_MIN_BATCH
                and (fork_mode == 1 or (fork_mode == 2) == torch.is_grad_enabled())):
            slot = self.__dict__.get("_fork_slot")
            if slot is None:
                slot = mc.new_fork_slot()
                object.__setattr__(self, "_fork_slot", slot)      # (replicas made by nn.DataParallel inherit it: same slot, their own device)
        chain = (slot, private)
        skip = self.to_rgb1(out, latent[:, 1], _s=bs(1), _chain=chain)
        rgb_img_list = [skip]
        i = 1
        for blk, to_rgb in enumerate(self.to_rgbs):
            for j in (0, 1):
                out = self.convs[2 * blk + j](out, latent[:, i + j], rss, noise=noise[2 * blk + 1 + j], _s=bs(2 + 3 * blk + j),
                                              _prep=pp(1 + 2 * blk + j))
                if rss:
                    out, sc = out
                    styles_list.append(sc)
            if rss and (i + 3) == latent.shape[1]:   # only the last ToRGB reports its scalars (ref :637-639)
                skip, sc = to_rgb(out, latent[:, i + 2], skip, True, _s=bs(4 + 3 * blk), _chain=chain)
                styles_list.append(sc)
            else:
                skip = to_rgb(out, latent[:, i + 2], skip, _s=bs(4 + 3 * blk), _chain=chain)
            rgb_img_list.append(skip)
            i += 2
        mc.torgb_join(skip, slot)      # the caller's stream sees every RGB output
        image = skip
        if want_latent:
            return image, latent
        returns = rgb_img_list if return_rgb_list else image
        if rss:
            returns = returns, styles_list
        return returns


# ---------------------------------------------------------------------------------------------------
# Discriminator (reference model.py:670-798).  On the GPU every ConvLayer pattern the network uses is one fused op on
# libcagc: 3x3 + FusedLeakyReLU (Winograd / direct MFMA), Blur -> 3x3 stride 2 (pitched FIR + stride-2 MFMA conv),
# Blur -> 1x1 stride 2 skip (decimating FIR + 1x1 MFMA GEMM), from-RGB 1x1 + FusedLeakyReLU.
# ---------------------------------------------------------------------------------------------------
class ConvLayer(nn.Sequential):
    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)
        self._fused_down = downsample and kernel_size == 3 and list(blur_kernel) == [1, 3, 3, 1]
        self._fused_s1 = (not downsample) and kernel_size == 3 and activate and bias
        self._fused_1x1 = (not downsample) and kernel_size == 1 and activate and bias
        self._fused_skip = (downsample and kernel_size == 1 and not activate and not bias
                            and list(blur_kernel) == [1, 3, 3, 1])
        self._packed = None

    def _packed_weights(self, conv):
        w = conv.weight
        need_bwd = torch.is_grad_enabled()
        if w.requires_grad:
            return mc.pack_plain_weights(w, conv.scale, need_bwd)
        key = (w._version, w.data_ptr(), w.device)
        if self._packed is None or self._packed[0] != key or (need_bwd and self._packed[1][1] is None):
            self._packed = (key, mc.pack_plain_weights(w, conv.scale, need_bwd))
        return self._packed[1]

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._packed_wino = None
        self._packed_gemm = None
        return super()._apply(fn, *a, **k)

    def _gemm_weights(self, conv):
        """(A operand of scale*W, of its transpose) of a 1x1 conv, packed for the per-image GEMM kernel of the ResBlock skip
        (csrc/conv1x1.hip; a 1x1 convolution over NCHW is out[b] = W @ x[b]).  Cached for frozen weights like the other packed
        operands."""
        w = conv.weight
        def make():
            w2 = w.detach().reshape(w.shape[0], w.shape[1])
            return mc.pack_gemm1x1(w2, conv.scale, False), mc.pack_gemm1x1(w2, conv.scale, True)
        if w.requires_grad:
            return make()
        key = (w._version, w.data_ptr(), w.device)
        c = getattr(self, "_packed_gemm", None)
        if c is None or c[0] != key:
            c = (key, make())
            self._packed_gemm = c
        return c[1]

    def _wino_weights(self, conv):
        w = conv.weight
        need_bwd = torch.is_grad_enabled()
        def bwd_pack():
            return mc.pack_wino(w, conv.scale, True) if mc.WINO_DGRAD else mc.pack_plain_weights(w, conv.scale, True)[1]
        if w.requires_grad:
            return mc.pack_wino(w, conv.scale, False), (bwd_pack() if need_bwd else None)
        key = (w._version, w.data_ptr(), w.device)
        c = getattr(self, "_packed_wino", None)   # own slot: the same layer may see Winograd-eligible and other sizes
        if c is None or c[0] != key or (need_bwd and c[1][1] is None):
            c = (key, (mc.pack_wino(w, conv.scale, False), bwd_pack() if need_bwd else None))
            self._packed_wino = c
        return c[1]

    def forward(self, input):
        # Every fused op below builds a differentiable backward when autograd asks for one (create_graph=True — the
        # reference's R1 step, train.py:194-200, runs on this class unchanged); `composed_autograd()` forces the composed
        # formulation, whose convolution is the closed ConvF/ConvD/ConvW family on the same kernels.
        if input.is_cuda and mc.composed_active():
            return super().forward(input)
        # 3x3 stride-1 conv + FusedLeakyReLU as one Winograd MFMA kernel (>= 32 px feature maps)
        if (self._fused_s1 and mc.use_hip(input) and input.dtype == torch.float32 and mc.wino_ok(input.shape[2], input.shape[3])
                and self[1].bias is not None and self[1].negative_slope == 0.2):
            conv, act = self[0], self[1]
            up_fwd, up_bwd = self._wino_weights(conv)
            return mc._Conv3x3Act.apply(input, conv.weight, act.bias, up_fwd, up_bwd, conv.scale)
        # from-RGB layer with frozen weights (generator step): two streaming kernels (3 input channels is no GEMM)
        if (self._fused_1x1 and mc.use_hip(input) and input.dtype == torch.float32 and input.shape[1] == 3
                and (input.shape[2] * input.shape[3]) % 4 == 0 and self[1].bias is not None and self[1].negative_slope == 0.2
                and not self[0].weight.requires_grad and not self[1].bias.requires_grad and not mc.composed_active()):
            return mc._FromRGBFrozen.apply(input, self[0].weight, self[1].bias, self[0].scale)
        # 1x1 conv + FusedLeakyReLU (the discriminator's from-RGB layer), and the 3x3 ones too small for the Winograd tiling
        # (4^2 .. 16^2), as one implicit-GEMM launch with the bias + LeakyReLU epilogue
        if ((self._fused_1x1 or self._fused_s1) and mc.use_hip(input) and input.dtype == torch.float32
                and input.shape[3] % 4 == 0 and self[1].bias is not None and self[1].negative_slope == 0.2):
            conv, act = self[0], self[1]
            wp_fwd, wp_bwd = self._packed_weights(conv)
            return mc._ConvActDirect.apply(input, conv.weight, act.bias, wp_fwd, wp_bwd, conv.scale)
        # Blur -> 3x3 stride-2 conv as one op on the hand-written MFMA kernel (odd blurred size 2*Ho+1)
        if (self._fused_down and mc.use_hip(input) and input.dtype == torch.float32
                and (input.shape[2] + sum(self[0].pad) - 3) % 2 == 1 and (input.shape[3] + sum(self[0].pad) - 3) % 2 == 1):
            blur, conv = self[0], self[1]
            wp_fwd, wp_bwd = self._packed_weights(conv)
            out = mc._BlurConvS2.apply(input, conv.weight, blur.kernel, wp_fwd, wp_bwd, tuple(blur.pad), conv.scale)
            if conv.bias is not None:
                out = out + conv.bias.view(1, -1, 1, 1)
            for layer in list(self)[2:]:
                out = layer(out)
            return out
        # Blur -> 1x1 stride-2 conv (ResBlock skip): FIR evaluated at the kept positions only, 1x1 conv as NCHW MFMA GEMM
        if (self._fused_skip and mc.use_hip(input) and input.dtype == torch.float32 and input.shape[2] % 2 == 0
                and input.shape[3] % 2 == 0 and self[1].bias is None):
            blur, conv = self[0], self[1]
            wp_fwd, wp_bwd = self._packed_weights(conv)
            return mc._BlurDownConv1x1.apply(input, conv.weight, blur.kernel, wp_fwd, wp_bwd, tuple(blur.pad), conv.scale)
        return super().forward(input)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def _frozen_fast_path(self, input):
        """Generator step (D frozen, only the input gradient wanted), Winograd-sized feature map, the stock layer pattern."""
        c1, c2, sk = self.conv1, self.conv2, self.skip
        if not (mc.FUSE_RESBLOCK and mc.WINO_DGRAD and mc.FUSE_ACT_DGRAD and mc.use_hip(input) and input.dtype == torch.float32
                and torch.is_grad_enabled() and input.requires_grad and input.dim() == 4 and not mc.composed_active()):
            return False        # (second-order passes announce themselves with mc.composed_autograd(): layer-by-layer, differentiable)
        if any(p.requires_grad for p in self.parameters()):
            return False
        H, W = input.shape[2], input.shape[3]
        return (c1._fused_s1 and c2._fused_down and sk._fused_skip and mc.wino_ok(H, W) and H % 2 == 0 and W % 2 == 0
                and c1[1].bias is not None and c1[1].negative_slope == 0.2 and c2[1].bias is None and len(c2) == 3
                and isinstance(c2[2], FusedLeakyReLU) and c2[2].bias is not None and c2[2].negative_slope == 0.2
                and tuple(c2[0].pad) == (2, 2) and tuple(sk[0].pad) == (1, 1) and sk[1].bias is None)

    def forward(self, input):
        if self._frozen_fast_path(input):
            c1, c2, sk = self.conv1, self.conv2, self.skip
            up1_fwd, up1_bwd = c1._wino_weights(c1[0])
            wp2_fwd, wp2_bwd = c2._packed_weights(c2[1])
            wpsk_fwd, wpsk_bwd = sk._gemm_weights(sk[1])
            return mc._ResBlockFrozen.apply(input, c1[0].weight, c1[1].bias, up1_fwd, up1_bwd, c2[1].weight, c2[2].bias, wp2_fwd,
                                            wp2_bwd, c2[0].kernel, tuple(c2[0].pad), sk[1].weight, wpsk_fwd, wpsk_bwd,
                                            sk[0].kernel, tuple(sk[0].pad))
        a, b = self.conv2(self.conv1(input)), self.skip(input)
        if mc.use_hip(a):
            return mc.add_scale(a, b, 1.0 / math.sqrt(2))
        return (a + b) / math.sqrt(2)


class Discriminator(nn.Module):
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                    256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        convs = [ConvLayer(3, channels[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"),
                                          EqualLinear(channels[4], 1))

    def forward(self, input):
        out = self.convs(input)
        batch, channel, height, width = out.shape
        group = min(batch, self.stddev_group)
        stddev = out.view(group, -1, self.stddev_feat, channel // self.stddev_feat, height, width)
        stddev = torch.sqrt(stddev.var(0, unbiased=False) + 1e-8)
        stddev = stddev.mean([2, 3, 4], keepdims=True).squeeze(2)
        stddev = stddev.repeat(group, 1, height, width)
        out = torch.cat([out, stddev], 1)
        out = self.final_conv(out)
        return self.final_linear(out.view(batch, -1))
