"""Oracle restatement of the reference model layer (L2): Generator / Discriminator forward as pure
functions of a state dict — torch fp32, CPU.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import math

import torch
import torch.nn.functional as F

from .ref_ops import (SQRT2, equal_linear_ref, fir_kernel, fused_leaky_relu_ref, modulated_conv2d_ref,
                      pixel_norm_ref, upfirdn2d_ref)


def _n_style_layers(sd):
    return len({k.split(".")[1] for k in sd if k.startswith("style.")})


def mapping_ref(sd, z, lr_mlp=0.01):
    """reference model.py:421-430 — PixelNorm then n_mlp EqualLinear(lr_mul=.01, fused lrelu)."""
    h = pixel_norm_ref(z)
    for i in range(1, _n_style_layers(sd) + 1):
        h = equal_linear_ref(h, sd[f"style.{i}.weight"], sd[f"style.{i}.bias"], lr_mul=lr_mlp, activation=True)
    return h


def _styled_conv(sd, prefix, x, w, noise, upsample):
    """reference model.py:351-367: modconv -> noise injection (:298-303) -> fused lrelu."""
    y, s = modulated_conv2d_ref(x, w, sd[prefix + ".conv.weight"], sd[prefix + ".conv.modulation.weight"],
                                sd[prefix + ".conv.modulation.bias"], upsample=upsample)
    if noise is None:
        noise = torch.randn(y.shape[0], 1, y.shape[2], y.shape[3], dtype=y.dtype)
    y = y + sd[prefix + ".noise.weight"] * noise
    return fused_leaky_relu_ref(y, sd[prefix + ".activate.bias"]), s


def _to_rgb(sd, prefix, x, w, skip):
    """reference model.py:380-395: 1x1 modconv (no demod) + bias + upfirdn2d(skip, up=2, pad=(2,1))."""
    y, s = modulated_conv2d_ref(x, w, sd[prefix + ".conv.weight"], sd[prefix + ".conv.modulation.weight"],
                                sd[prefix + ".conv.modulation.bias"], demodulate=False)
    y = y + sd[prefix + ".bias"]
    if skip is not None:
        y = y + upfirdn2d_ref(skip, fir_kernel([1, 3, 3, 1], 4.0), up=2, pad=(2, 1))
    return y, s


def generator_forward_ref(sd, zs=None, latents=None, inject_index=None, noise=None, randomize_noise=True,
                          truncation=1.0, truncation_latent=None, return_rgb_list=False,
                          return_style_scalars=False, return_latent=False):
    """reference model.py:545-659.  `zs`: list of 1 or 2 [B,style_dim] z (mapped here) or
    `latents`: list of already-mapped w (`input_is_latent`).  `noise`: list of per-layer tensors,
    None -> fresh N(0,1) (`randomize_noise=True`) or the `noises.noise_i` buffers."""
    n_conv = len([k for k in sd if k.startswith("convs.") and k.endswith(".conv.weight")])
    num_layers = n_conv + 1
    n_latent = num_layers + 1                      # == log2(size)*2 - 2 (:521)
    styles = [mapping_ref(sd, z) for z in zs] if latents is None else list(latents)
    if noise is None:
        noise = [None] * num_layers if randomize_noise else [sd[f"noises.noise_{i}"] for i in range(num_layers)]
    if truncation < 1:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    else:
        assert inject_index is not None, "oracle wants the mixing index explicit"
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    b = latent.shape[0]
    x = sd["input.input"].repeat(b, 1, 1, 1)                                             # :316-320
    scal = []
    x, s = _styled_conv(sd, "conv1", x, latent[:, 0], noise[0], upsample=False)
    scal.append(s)
    skip, _ = _to_rgb(sd, "to_rgb1", x, latent[:, 1], None)
    rgbs = [skip]
    i = 1
    for blk in range(n_conv // 2):
        x, s = _styled_conv(sd, f"convs.{2 * blk}", x, latent[:, i], noise[2 * blk + 1], upsample=True)
        scal.append(s)
        x, s = _styled_conv(sd, f"convs.{2 * blk + 1}", x, latent[:, i + 1], noise[2 * blk + 2], upsample=False)
        scal.append(s)
        skip, s = _to_rgb(sd, f"to_rgbs.{blk}", x, latent[:, i + 2], skip)
        if i + 3 == n_latent:                      # only the last ToRGB reports its scalars (:637-639)
            scal.append(s)
        rgbs.append(skip)
        i += 2
    out = rgbs if return_rgb_list else rgbs[-1]
    ret = (out,)
    if return_style_scalars:
        ret += (scal,)
    if return_latent:
        ret += (latent,)
    return ret[0] if len(ret) == 1 else ret


def path_lengths_ref(image, latent, pl_noise):
    """reference model.py:661-666 (pl_noise = randn_like(image), un-normalised)."""
    n = pl_noise / math.sqrt(image.shape[2] * image.shape[3])
    (g,) = torch.autograd.grad((image * n).sum(), latent, create_graph=True)
    return torch.sqrt(g.pow(2).sum(2).mean(1))


# ------------------------------------------------------------------------------------------------
# Discriminator (reference model.py:670-798)
# ------------------------------------------------------------------------------------------------
def _equal_conv(x, w, bias=None, stride=1, padding=0):
    """reference model.py:119-128."""
    return F.conv2d(x, w * (1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])), bias=bias, stride=stride,
                    padding=padding)


def _conv_layer(sd, prefix, x, k, downsample=False, activate=True, bias=True):
    """reference model.py:670-716: [Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU]."""
    idx = 0
    if downsample:
        p = (4 - 2) + (k - 1)
        x = upfirdn2d_ref(x, fir_kernel([1, 3, 3, 1]), pad=((p + 1) // 2, p // 2))
        idx = 1
    w = sd[f"{prefix}.{idx}.weight"]
    cb = sd.get(f"{prefix}.{idx}.bias") if (bias and not activate) else None
    x = _equal_conv(x, w, cb, stride=2 if downsample else 1, padding=0 if downsample else k // 2)
    if activate:
        if bias:
            x = fused_leaky_relu_ref(x, sd[f"{prefix}.{idx + 1}.bias"])
        else:
            x = F.leaky_relu(x, 0.2) * SQRT2
    return x


def discriminator_forward_ref(sd, x):
    """reference model.py:780-798."""
    n_res = len({k.split(".")[1] for k in sd if k.startswith("convs.")}) - 1
    h = _conv_layer(sd, "convs.0", x, 1)
    for r in range(1, n_res + 1):
        a = _conv_layer(sd, f"convs.{r}.conv1", h, 3)
        a = _conv_layer(sd, f"convs.{r}.conv2", a, 3, downsample=True)
        sk = _conv_layer(sd, f"convs.{r}.skip", h, 1, downsample=True, activate=False, bias=False)
        h = (a + sk) / SQRT2
    b, c, hh, ww = h.shape
    group = min(b, 4)
    sdv = h.reshape(group, -1, 1, c, hh, ww)
    sdv = torch.sqrt(sdv.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdim=True).squeeze(2)
    h = torch.cat([h, sdv.repeat(group, 1, hh, ww)], 1)
    h = _conv_layer(sd, "final_conv", h, 3)
    h = equal_linear_ref(h.reshape(b, -1), sd["final_linear.0.weight"], sd["final_linear.0.bias"], activation=True)
    return equal_linear_ref(h, sd["final_linear.1.weight"], sd["final_linear.1.bias"])


def regenerate_state_dict(keys_shapes, seed):
    """Re-draw a randomly initialised reference module's state dict from its seed, walking the keys in
    registration order: `weight` ~ randn (EqualConv2d model.py:105-107, EqualLinear :143 with lr_mul=1),
    `bias` = 0, `kernel` = the [1,3,3,1] FIR.  Used for the 29 MB Discriminator, which is too large to
    commit; the fixtures carry per-tensor checksums to prove the regeneration is exact."""
    g = torch.Generator().manual_seed(int(seed))
    sd = {}
    for key, shape in keys_shapes:
        if key.endswith("weight"):
            sd[key] = torch.randn(*shape, generator=g)
        elif key.endswith("kernel"):
            sd[key] = fir_kernel([1, 3, 3, 1])
        else:
            sd[key] = torch.zeros(*shape)
    return sd


def regenerate_generator_state_dict(keys_shapes, seed, lr_mlp=0.01):
    """A seeded recipe for a Generator's parameters and noise buffers, walking the keys in registration order (FIR `kernel`
    buffers are skipped: the constructor builds them, and the fixture's checksums pin them).  Used for the style_dim = 512 tiny
    generator (`tests/golden/generator512*`): its mapping network alone is 2 MB, too large to commit, so gen_golden.py loads THIS
    recipe into the reference's Generator and the tests load it into the product's / hand it to the oracle.  Values: mapping
    weights ~ N(0,1) / lr_mlp (as EqualLinear initialises them, reference model.py:143), modulation bias 1 + 0.1 N, noise weights
    0.1 + 0.05 N (the reference's zero init would hide the noise path), other biases 0.1 N, everything else N(0,1)."""
    g = torch.Generator().manual_seed(int(seed))
    sd = {}
    for key, shape in keys_shapes:
        if key.endswith("kernel"):
            continue
        t = torch.randn(*shape, generator=g)
        if key.endswith("modulation.bias"):
            t = 1.0 + 0.1 * t
        elif key.endswith("noise.weight"):
            t = 0.1 + 0.05 * t
        elif key.startswith("style.") and key.endswith("weight"):
            t = t / lr_mlp
        elif key.endswith("bias"):
            t = 0.1 * t
        sd[key] = t
    return sd
