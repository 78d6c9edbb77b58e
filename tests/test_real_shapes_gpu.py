"""Real-size shapes of BASELINE configs[1] (256 px, global batch 16) against float64 / through size-independent properties:
the discriminator's conv layers at their actual channel counts and resolutions (forward, data gradient, weight and bias
gradient — the shapes whose weight-gradient / launch plans tiny goldens never select), Discriminator(256) at batch 16, and
one whole configs[1] KD generator step at batch 16.  Reference: model.py:670-798, train.py:280-308."""
import os

import pytest
import torch
from torch.nn import functional as F

import cagc.model as M
from cagc import kd
from _util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


# (cin, cout, H, downsample, B): ResBlock conv1 / conv2 of Discriminator(256) (model.py:756-778)
D_LAYERS = [(128, 128, 256, False, 1), (128, 256, 256, True, 1), (256, 256, 128, False, 1), (256, 512, 128, True, 1),
            (512, 512, 64, False, 1), (512, 512, 64, True, 2), (512, 512, 32, False, 2), (512, 512, 32, True, 2), (512, 512, 16, False, 4)]


@pytest.mark.parametrize("cfg", D_LAYERS)
def test_discriminator_conv_layers_real_shapes_vs_float64(cfg, wino4_policy):
    """ConvLayer = [Blur ->] EqualConv2d -> FusedLeakyReLU on the HIP kernels (Winograd / register-direct stride-2 conv,
    register-direct weight gradient incl. the phase-planar role swap) vs the same layer evaluated in float64: output,
    input gradient, weight gradient, bias gradient."""
    cin, cout, H, down, B = cfg
    torch.manual_seed(31)
    layer = M.ConvLayer(cin, cout, 3, downsample=down)
    with torch.no_grad():
        layer[-1].bias.copy_(0.1 * torch.randn(cout))
    x = torch.randn(B, cin, H, H)
    conv = layer[1] if down else layer[0]
    # float64 reference of the same layer
    w64 = conv.weight.detach().double().requires_grad_(True)
    b64 = layer[-1].bias.detach().double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    h = x64
    if down:
        k = layer[0].kernel.double()
        p = layer[0].pad
        h = F.conv2d(F.pad(h, (p[0], p[1], p[0], p[1])).reshape(-1, 1, H + p[0] + p[1], H + p[0] + p[1]),
                     torch.flip(k, [0, 1])[None, None]).reshape(B, cin, H + p[0] + p[1] - 3, H + p[0] + p[1] - 3)
    pre = F.conv2d(h, w64 * conv.scale, stride=2 if down else 1, padding=0 if down else 1) + b64[None, :, None, None]
    lg = layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    yg = lg(xg)
    # common-gate protocol: LeakyReLU gates where fp32 and float64 pick different sides must sit at rounding level
    # (|pre-activation| < 1e-5 of the layer scale) and be few; the float64 layer is then evaluated on the HIP run's gate pattern
    gate = (yg.detach() > 0).cpu()
    dis = gate != (pre.detach() > 0)
    # stride-1 layers of D run on the Winograd F(4x4,3x3) kernel (cout, cin >= 128): per-layer bar 5e-5 (observed 1-2e-5), F(2x2) /
    # register-direct layers 5e-6
    f4 = (not down) and os.environ.get("CAGC_WINO_F4", "1") != "0" and cout >= 128 and cin >= 128
    bar = 5e-5 if f4 else 5e-6
    if int(dis.sum()):
        assert float(pre.detach()[dis].abs().max()) < (1e-4 if f4 else 1e-5) * float(pre.detach().abs().max()), "gate flip above rounding level"
        assert int(dis.sum()) <= max(4, (1e-4 if f4 else 1e-5) * dis.numel()), int(dis.sum())
    y = torch.where(gate, pre, 0.2 * pre) * 2 ** 0.5
    go = torch.randn(y.shape)
    gx64, gw64, gb64 = torch.autograd.grad(y, [x64, w64, b64], go.double())
    gxg, gwg, gbg = torch.autograd.grad(yg, [xg, (lg[1] if down else lg[0]).weight, lg[-1].bias], go.to(DEV))
    assert _rel(yg, y) <= bar, ("out", cfg, _rel(yg, y))
    assert _rel(gxg, gx64) <= bar, ("grad x", cfg, _rel(gxg, gx64))
    assert _rel(gwg, gw64) <= max(bar, 2e-5), ("grad weight", cfg, _rel(gwg, gw64))     # sums over B*H*W pixels of random products
    assert _rel(gbg, gb64) <= max(bar, 2e-5), ("grad bias", cfg, _rel(gbg, gb64))


def test_discriminator_256_batch16_properties():
    """Discriminator(256) at batch 16 (frozen, as on the generator step): finite; the minibatch-stddev groups are the strided
    sample sets {n, n+4, n+8, n+12} (model.py:784-790), so scores of a group depend on that group's images only; the input
    gradient is linear in the upstream gradient."""
    torch.manual_seed(32)
    d = M.Discriminator(256).to(DEV)
    kd.requires_grad(d, False)
    x = torch.randn(16, 3, 256, 256, device=DEV, requires_grad=True)
    y = d(x)
    assert tuple(y.shape) == (16, 1) and torch.isfinite(y).all()
    with torch.no_grad():
        for n in range(4):
            yn = d(x[n::4].detach())
            assert_close(yn, y[n::4].detach(), 2e-5, f"stddev group {n}")
    u = torch.randn(16, 1, device=DEV)
    (g1,) = torch.autograd.grad(y, x, u, retain_graph=True)
    (g2,) = torch.autograd.grad(y, x, -3 * u)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert_close(g2, -3 * g1, 1e-5, "dgrad linearity")


def test_kd_step_256_batch16_properties():
    """One whole configs[1] KD generator step (pruned student [154x10,77,77,39,39] + full teacher + Discriminator(256), batch
    16): losses and every gradient finite; backward additive over the two loss terms; images batch independent; the step runs
    twice to the same gradients (no fp32 atomics left on the convolution path at this size: run-to-run deviation is bounded at
    rounding level, not at the parity bar's)."""
    student, teacher, disc = kd.build_synthetic_workload(256, DEV, seed=0)
    step = kd.KDStep(student, teacher, disc)
    B = 16
    gen = torch.Generator(device=DEV).manual_seed(6)
    zs = [torch.randn(B, 512, device=DEV, generator=gen), torch.randn(B, 512, device=DEV, generator=gen)]
    nl = student.num_layers
    sn = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV, generator=gen) for i in range(nl)]
    tn = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV, generator=gen) for i in range(nl)]
    mask = kd.ellipse_mask(B, 256, DEV)
    kd.requires_grad(student, True)
    kd.requires_grad(disc, False)
    params = [p for p in student.parameters()]
    names = [n for n, _ in student.named_parameters()]

    def grads(wg, wk):
        g_loss, kd_l1, img = step.g_losses(zs, 5, mask, sn, tn)
        gs = torch.autograd.grad(wg * g_loss + wk * kd_l1, params, allow_unused=True)
        return g_loss.detach(), kd_l1.detach(), img.detach(), gs

    from cagc import _lib
    lib = _lib.load()
    # Deterministic mode (CAGC_DETERMINISTIC=1 / cagc_set_tuning): no fp32-atomic K split in the convolution kernels, so every
    # forward pass is bit-reproducible.  The LeakyReLU gates of the 10^8 activations of student and discriminator are then the
    # same in every run, and gradients repeat to summation-order rounding instead of to the effect of flipped gates (default
    # mode, measured on this step: 2.3e-3 of a tensor's scale)
    prev_det = _lib.set_tuning("deterministic", 1)
    try:
        gl, kl, img, g_all = grads(1.0, 1.0)
        assert torch.isfinite(gl) and torch.isfinite(kl) and kl.item() > 0
        gl2, kl2, img2, g_again = grads(1.0, 1.0)
        assert torch.equal(img2, img), "deterministic mode: the forward pass is bit-reproducible"
        assert torch.equal(gl2, gl) and torch.equal(kl2, kl), "deterministic mode: losses not bit-reproducible"
        # ... and so is every gradient: the backward's many-to-one reductions (grad-bias, styled-epilogue and style sums,
        # ToRGB weight sums, the L1 loss) run through the order-independent fixed-point sink (csrc/common.h DetSink)
        differing = []
        for n, a, b in zip(names, g_all, g_again):
            if a is not None:
                assert torch.isfinite(a).all(), n
                if not torch.equal(a, b):
                    differing.append((n, _rel(a, b)))
        assert not differing, f"deterministic mode: gradients not bit-identical between two runs: {differing[:8]}"
        _, _, _, g_g = grads(1.0, 0.0)
        _, _, _, g_k = grads(0.0, 1.0)
        for n, a, b, c in zip(names, g_all, g_g, g_k):
            if a is not None:
                assert_close(a, b + c, 2e-5 if a.numel() > 1 else 1e-3, "additivity " + n)
    finally:
        _lib.set_tuning("deterministic", prev_det)
    with torch.no_grad():
        one = student([z[3:4] for z in zs], inject_index=5, noise=[n[3:4] for n in sn])
        assert_close(one, img[3:4], 1e-5, "student batch independence")
    losses = step.g_step(zs, 5, mask, sn, tn)
    assert all(torch.isfinite(v) for v in losses.values()) and all(torch.isfinite(p).all() for p in student.parameters())
