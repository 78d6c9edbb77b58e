"""-m gpu: the HIP path (through the C ABI of libcagc_hip.so) against (1) the golden vectors captured from the
reference CPU path and (2) the oracle (oracle/, torch-fp32 CPU restatement) on seeded inputs at sizes the oracle
finishes in seconds.  Parity bar: max|a-b| / max|b| <= 1e-3 (BASELINE.json north_star), far
tighter in practice because the MFMA path is an exact fp32 FMA chain."""
import pytest
import torch

import cagc.model as M
from cagc import _lib, kd
from cagc.op import fused_leaky_relu, upfirdn2d
from oracle import ref_kd, ref_model, ref_ops
from _util import assert_close, assert_grad_matches_sample, load_json, load_npz, sub

pytestmark = pytest.mark.gpu
TOL = 1e-3
TIGHT = 2e-5


def wino_f4(k_ch, m_ch, H, W, up=False):
    """The layer's 3x3 conv (GEMM K = k_ch, M = m_ch) runs on the Winograd F(4x4,3x3) kernel (csrc/conv_wino4.hip: M >= 128,
    K >= 128, Winograd-eligible size, CAGC_WINO_F4 != 0) — its transforms cost ~1.5 digits against F(2x2): the per-layer float64
    bar is 5e-5 there (observed 1-2.2e-5), 5e-6 elsewhere (observed 2-7e-7)."""
    import os
    return (not up and os.environ.get("CAGC_WINO_F4", "1") != "0" and m_ch >= 128 and k_ch >= 128 and H % 8 == 0 and W % 32 == 0)
DEV = "cuda"


def cu(t):
    return t.to(DEV)


def test_library_loaded_and_gfx950():
    assert _lib.load().cagc_arch() == b"gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_fused_leaky_relu_golden_first_and_second_order():
    g = load_npz("fused_act")
    for k in ("2d_b", "4d_b", "2d_nb", "4d_nb"):
        x = cu(g[k + "_x"]).requires_grad_(True)
        b = cu(g[k + "_b"]).requires_grad_(True) if (k + "_b") in g else None
        go = cu(g[k + "_go"]).requires_grad_(True)
        y = fused_leaky_relu(x, b)
        assert_close(y, g[k + "_y"], TIGHT, k + " y")
        ins = [x] + ([b] if b is not None else [])
        grads = torch.autograd.grad(y, ins, go, create_graph=True)
        assert_close(grads[0], g[k + "_gx"], TIGHT, k + " gx")
        if b is not None:
            assert_close(grads[1], g[k + "_gb"], TIGHT, k + " gb")
        (ggo,) = torch.autograd.grad(grads[0], go, cu(g[k + "_ggi"]))
        assert_close(ggo, g[k + "_ggo"], TIGHT, k + " ggo")


@pytest.mark.parametrize("shape", [(16, 39, 256, 256), (4, 154, 16, 16), (3, 7, 33, 17), (8, 512)])
def test_fused_leaky_relu_large_vs_oracle(shape):
    torch.manual_seed(1)
    x = torch.randn(*shape)
    b = torch.randn(shape[1])
    go = torch.randn(*shape)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_ops.fused_leaky_relu_ref(xr, br)
    gr = torch.autograd.grad(yr, [xr, br], go)
    xg, bg = cu(x).requires_grad_(True), cu(b).requires_grad_(True)
    yg = fused_leaky_relu(xg, bg)
    gg = torch.autograd.grad(yg, [xg, bg], cu(go))
    assert torch.equal(yg.cpu(), yr.detach()), "elementwise op must be bit-exact"
    assert_close(gg[0], gr[0], 1e-6, "grad_input")   # gout*(gate*scale) vs (gout*scale)*gate: 1-ulp association
    assert_close(gg[1], gr[1], 1e-4, "grad_bias (reduction order differs)")


def test_upfirdn2d_golden_all_configs_with_grad_and_gradgrad():
    g = load_npz("upfirdn2d")
    for c in load_json("upfirdn2d_cases"):
        n = c["name"]
        x = cu(g[n + "_x"]).requires_grad_(True)
        k = cu(g[n + "_k"])
        y = upfirdn2d(x, k, up=c["up"], down=c["down"], pad=tuple(c["pad"]))
        assert_close(y, g[n + "_y"], TIGHT, n + " y")
        go = cu(g[n + "_go"]).requires_grad_(True)
        (gx,) = torch.autograd.grad(y, x, go, create_graph=True)
        assert_close(gx, g[n + "_gx"], TIGHT, n + " gx")
        # double backward: d(gx)/d(go) contracted with v == forward op applied to v (linear op)
        v = torch.randn_like(gx)
        (ggo,) = torch.autograd.grad(gx, go, v)
        assert_close(ggo, ref_ops.upfirdn2d_ref(v.cpu(), g[n + "_k"], up=c["up"], down=c["down"], pad=tuple(c["pad"])), TIGHT,
                     n + " gradgrad")


@pytest.mark.parametrize("cfg", [((16, 128, 257, 257), 4.0, 1, 1, (1, 1)), ((4, 128, 128, 128), 1.0, 1, 1, (2, 2)),
                                 ((16, 3, 128, 128), 4.0, 2, 1, (2, 1)), ((16, 3, 256, 256), 1.0, 1, 2, (1, 1)),
                                 ((2, 5, 67, 41), 1.0, 1, 1, (1, 1)), ((4, 128, 128, 128), 1.0, 1, 2, (1, 1)),
                                 ((4, 256, 64, 64), 4.0, 2, 1, (2, 1)), ((3, 5, 33, 47), 1.0, 1, 2, (1, 1)),
                                 ((3, 5, 33, 47), 4.0, 2, 1, (2, 1)), ((2, 7, 34, 46), 1.0, 1, 2, (2, 1))])
def test_upfirdn2d_full_size_vs_oracle(cfg):
    shape, gain, up, down, pad = cfg
    torch.manual_seed(2)
    x = torch.randn(*shape)
    k = ref_ops.fir_kernel([1, 3, 3, 1], gain)
    y = upfirdn2d(cu(x), cu(k), up=up, down=down, pad=pad)
    assert_close(y, ref_ops.upfirdn2d_ref(x, k, up=up, down=down, pad=pad), TIGHT, str(cfg))


def _modconv_module(c, g, n):
    m = M.ModulatedConv2d(c["cin"], c["cout"], c["k"], c["style_dim"], demodulate=c.get("demodulate", True),
                          upsample=c.get("upsample", False), downsample=c.get("downsample", False))
    with torch.no_grad():
        m.weight.copy_(g[n + "_weight"])
        m.modulation.weight.copy_(g[n + "_mod_weight"])
        m.modulation.bias.copy_(g[n + "_mod_bias"])
    return m.to(DEV)


def test_modulated_conv_golden_plain_up_rgb_all_grads():
    g = load_npz("modconv")
    for c in load_json("modconv_cases"):
        n = c["name"]
        m = _modconv_module(c, g, n)
        x = cu(g[n + "_x"]).requires_grad_(True)
        w = cu(g[n + "_w"]).requires_grad_(True)
        y, s = m(x, w, return_style_scalars=True)
        assert_close(y, g[n + "_y"], TOL, n + " y")
        assert_close(s, g[n + "_s"], TIGHT, n + " s")
        grads = torch.autograd.grad(y, [x, w, m.weight, m.modulation.weight, m.modulation.bias], cu(g[n + "_go"]))
        for name, gr in zip(("x", "w", "weight", "mod_weight", "mod_bias"), grads):
            assert_close(gr, g[f"{n}_g{name}"], TOL, f"{n} g{name}")


@pytest.mark.parametrize("cfg", [  # (B, cin, cout, H, W, upsample)
    (2, 154, 154, 16, 16, False), (2, 154, 77, 32, 32, True), (2, 77, 39, 64, 64, True), (2, 39, 39, 96, 64, False),
    (1, 512, 512, 4, 4, False), (3, 512, 256, 8, 8, True), (2, 128, 128, 64, 64, False), (5, 20, 10, 12, 20, False),
    (16, 154, 154, 4, 4, True), (2, 24, 20, 9, 7, True), (1, 16, 16, 6, 10, True)])   # last two: odd / non-pow2 sizes (scalar fallbacks)
def test_styled_conv_vs_oracle_forward_and_all_grads(cfg):
    B, cin, cout, H, W, up = cfg
    torch.manual_seed(3)
    m = M.StyledConv(cin, cout, 3, 64, upsample=up)
    with torch.no_grad():
        m.noise.weight.fill_(0.3)
        m.activate.bias.copy_(0.2 * torch.randn(cout))
        m.conv.modulation.bias.add_(0.3 * torch.randn(cin))
    sd = {"c." + k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    w = torch.randn(B, 64)
    oh, ow = (2 * H, 2 * W) if up else (H, W)
    noise = torch.randn(B, 1, oh, ow)
    go = torch.randn(B, cout, oh, ow)
    # oracle
    names = ["c.conv.weight", "c.conv.modulation.weight", "c.conv.modulation.bias", "c.noise.weight", "c.activate.bias"]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sdr = dict(sd)
    sdr.update(leaves)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    # HIP
    mg = m.to(DEV)
    xg, wg = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    yg = mg(xg, wg, noise=cu(noise))
    # oracle on the HIP run's LeakyReLU gate pattern (a pre-activation within rounding of 0 may land on either side;
    # every such gate is checked to be at rounding level of the layer scale)
    gpu_gate = [(yg.detach() > 0).cpu()]
    with ref_ops.gates() as rec:
        with torch.no_grad():
            ref_model._styled_conv(sd, "c", x, w, noise, up)
    ref_ops.gate_disagreements(rec, gpu_gate, rounding=1e-4)
    with ref_ops.gates(force=gpu_gate):
        yr, _ = ref_model._styled_conv(sdr, "c", xr, wr, noise, up)
    gr = torch.autograd.grad(yr, [xr, wr] + [leaves[k] for k in names], go)
    assert_close(yg, yr, TOL, "out")
    params = dict(mg.named_parameters())
    gg = torch.autograd.grad(yg, [xg, wg] + [params[k[2:]] for k in names], cu(go))
    for nm, a, b in zip(["x", "style"] + names, gg, gr):
        assert_close(a, b, TOL if b.numel() > 1 else 3e-3, f"{cfg} grad {nm}")


@pytest.mark.parametrize("cfg", [(2, 39, 256, 256, True), (2, 154, 8, 8, True), (3, 154, 4, 4, False), (1, 7, 6, 10, False)])
def test_torgb_vs_oracle_forward_and_all_grads(cfg):
    B, C, H, W, has_skip = cfg
    torch.manual_seed(4)
    m = M.ToRGB(C, 64, upsample=has_skip)
    with torch.no_grad():
        m.bias.copy_(0.1 * torch.randn(1, 3, 1, 1))
    sd = {"r." + k: v.detach().clone() for k, v in m.state_dict().items()}
    x, w = torch.randn(B, C, H, W), torch.randn(B, 64)
    skip = torch.randn(B, 3, H // 2, W // 2) if has_skip else None
    go = torch.randn(B, 3, H, W)
    names = ["r.conv.weight", "r.conv.modulation.weight", "r.conv.modulation.bias", "r.bias"]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sdr = dict(sd)
    sdr.update(leaves)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    sr = skip.clone().requires_grad_(True) if has_skip else None
    yr, _ = ref_model._to_rgb(sdr, "r", xr, wr, sr)
    ins_r = [xr, wr] + [leaves[k] for k in names] + ([sr] if has_skip else [])
    gr = torch.autograd.grad(yr, ins_r, go)
    mg = m.to(DEV)
    xg, wg = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    sg = cu(skip).requires_grad_(True) if has_skip else None
    yg = mg(xg, wg, sg)
    assert_close(yg, yr, TOL, "rgb")
    params = dict(mg.named_parameters())
    ins_g = [xg, wg] + [params[k[2:]] for k in names] + ([sg] if has_skip else [])
    gg = torch.autograd.grad(yg, ins_g, cu(go))
    for nm, a, b in zip(["x", "style"] + names + ["skip"], gg, gr):
        assert_close(a, b, TOL, f"{cfg} grad {nm}")


def _tiny(g, meta):
    net = M.Generator(meta["config"]["size"], meta["config"]["style_dim"], meta["config"]["n_mlp"],
                      generator_net_shape=meta["config"]["shape"])
    net.load_state_dict(sub(g, "sd/"), strict=True)
    return net.to(DEV)


def test_tiny_generator_golden_image_rgbs_styles_grads_pathlength():
    g = load_npz("generator_tiny")
    meta = load_json("generator_tiny_keys")
    net = _tiny(g, meta)
    rgbs, scal = net([cu(g["z0"])], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    for i, r in enumerate(rgbs):
        assert_close(r, g[f"a_rgb{i}"], TOL, f"rgb{i}")
    for i, s in enumerate(scal):
        assert_close(s, g[f"a_style{i}"], TOL, f"style{i}")
    assert_close(net([cu(g["z0"]), cu(g["z1"])], inject_index=3, randomize_noise=False), g["c_img"], TOL, "mixing")
    assert_close(net(None, latent_styles=[cu(g["d_w0"])], input_is_latent=True, truncation=0.7,
                     truncation_latent=cu(g["d_mean_w"]), randomize_noise=False), g["d_img"], TOL, "truncation")
    noise = [cu(g[f"e_noise{i}"]) for i in range(meta["num_layers"])]
    assert_close(net([cu(g["z0"])], noise=noise), g["e_img"], TOL, "noise list")
    net.zero_grad()
    img = net([cu(g["z0"])], randomize_noise=False)
    assert_close(img, g["b_img"], TOL, "img")
    img.abs().mean().backward()
    params = dict(net.named_parameters())
    for k, v in sub(g, "b_grad/").items():
        gr = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        assert_close(gr, v, TOL if v.numel() > 1 else 5e-3, "grad " + k)
    # second order (composed twice-differentiable ops on the GPU, HIP upfirdn2d / fused act double-backward)
    from unittest import mock
    net.zero_grad()
    with mock.patch.object(torch, "randn_like", lambda t: cu(g["f_pl_noise"])):
        _, pl = net([cu(g["z0"])], PPL_regularize=True, randomize_noise=False)
    assert_close(pl, g["f_path_lengths"], TOL, "path lengths")
    (pl - 0.37).pow(2).mean().backward()
    for k, v in sub(g, "f_grad/").items():
        gr = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        assert_close(gr, v, 2e-3 if v.numel() > 1 else 1e-2, "pl grad " + k)


def test_generator512_golden_reaches_the_hip_linears(monkeypatch):
    """The reference-generated fixture at the REAL latent width (style_dim = 512, reference model.py:137-171,421-430): PixelNorm,
    the mapping network (csrc/mapping.hip), the modulation bank (csrc/modbank.hip) and _MixLatent all run on libcagc —
    CAGC_STRICT_HIP=1 turns any library GEMM / stock convolution into an error — against the reference's image, RGB list, style
    scalars, mapped latents and every parameter gradient (VERDICT r5 missing #2: every earlier golden used style_dim 24 / 32 / 64)."""
    monkeypatch.setenv("CAGC_STRICT_HIP", "1")
    g = load_npz("generator512")
    meta = load_json("generator512_keys")
    cfg = meta["config"]
    net = M.Generator(cfg["size"], cfg["style_dim"], cfg["n_mlp"], generator_net_shape=cfg["shape"])
    missing, unexpected = net.load_state_dict(ref_model.regenerate_generator_state_dict(meta["keys"], cfg["seed"]), strict=False)
    assert not unexpected and all(k.endswith("kernel") for k in missing)
    sd = net.state_dict()
    assert [k for k, _ in meta["keys"]] == list(sd.keys())
    for i, (k, v) in enumerate(sd.items()):      # incl. the FIR buffers the constructor built
        assert abs(float(v.double().sum()) - float(g["sd_checksum"][i])) <= 1e-6 * max(1.0, float(g["sd_abs_checksum"][i])), k
    net = net.to(DEV)
    assert_close(net.get_latent(cu(g["z0"])), g["a_w0"], TOL, "mapping")
    rgbs, scal = net([cu(g["z0"])], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    assert len(scal) == g["a_n_styles"]
    for i, r in enumerate(rgbs):
        assert_close(r, g[f"a_rgb{i}"], TOL, f"rgb{i}")
    for i, s_ in enumerate(scal):
        assert_close(s_, g[f"a_style{i}"], TOL, f"style{i}")
    net.zero_grad()
    img = net([cu(g["z0"]), cu(g["z1"])], inject_index=3, randomize_noise=False)
    assert_close(img, g["b_img"], TOL, "mixing image")
    img.abs().mean().backward()
    for k, p in net.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_grad_matches_sample(gr, g["b_grad/" + k], g["b_gsum/" + k], cfg, TOL if p.numel() > 1 else 5e-3, "grad " + k)
    # and with the device-side mixing index the graph-replayed step uses (_MixLatent)
    inj = torch.full((1,), 3, device=DEV, dtype=torch.long)
    assert_close(net([cu(g["z0"]), cu(g["z1"])], inject_index=inj, randomize_noise=False), g["b_img"], TOL, "mixing image (device index)")


def test_discriminator_golden():
    g = load_npz("discriminator32")
    d = M.Discriminator(32)
    d.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["seed"]), strict=True)
    d = d.to(DEV)
    x = cu(g["x"]).requires_grad_(True)
    y = d(x)
    assert_close(y, g["y"], TOL, "D out")
    (gx,) = torch.autograd.grad(torch.nn.functional.softplus(-y).mean(), x)
    # one LeakyReLU gate of 2,097,152 sits at a float64 pre-activation of 7e-7 and lands on the other side of 0 here
    # (test_discriminator_vs_float64_reference pins that: everything outside its 3x3 field agrees to the 1e-3 bar);
    # this fp32-golden comparison therefore keeps the looser whole-tensor bound
    assert_close(gx, g["gx"], 3e-3, "D input grad")


def _f64_bar(f64, key):
    """Parity bar against a float64 evaluation of the REFERENCE (tests/golden/float64_refs.npz): the north-star 1e-3, or
    three times the fp32 reference's own deviation from float64 where that is larger."""
    return max(TOL, 3.0 * float(f64[key]))


@pytest.mark.parametrize("wino_dgrad", [True, False])
def test_discriminator_vs_float64_reference(wino_dgrad, monkeypatch):
    """D(32) output and input gradient vs the REFERENCE evaluated in float64 (tests/golden/float64_refs.npz), with the
    stride-1 3x3 data gradients on the Winograd kernel and on the direct implicit GEMM.

    Measured (gpurun_out/run2.log, scripts/d32_diag.py): the whole deviation of the input gradient from the golden
    (1.34e-3, identical for both data-gradient kernels) comes from ONE LeakyReLU gate out of 2,097,152 in the first
    ResBlock whose float64 pre-activation is 7.3e-7 (activation scale 7.8): fp32 rounding puts it on the other side of
    0, and the error is confined to the 3x3 input positions that element reaches; every other position agrees to
    ~5e-7.  So the test (a) requires every gate disagreement to be at rounding level and few, and (b) holds the
    gradient to the 1e-3 bar everywhere, excluding only the receptive field of a disagreeing gate — where it still must
    stay within the magnitude one flipped gate can cause."""
    import copy
    from cagc.op import modconv as mc
    from _util import forward_with_activations, gate_flips, reach_mask
    monkeypatch.setattr(mc, "WINO_DGRAD", wino_dgrad)
    g, f64 = load_npz("discriminator32"), load_npz("float64_refs")
    d = M.Discriminator(32)
    d.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["seed"]), strict=True)
    d64 = copy.deepcopy(d).double()                      # the product's composed CPU path in float64 ...
    x64 = g["x"].double().requires_grad_(True)
    y64, outs64 = forward_with_activations(d64, x64)
    (gx64,) = torch.autograd.grad(torch.nn.functional.softplus(-y64).mean(), x64, retain_graph=True)
    assert_close(y64, f64["d32/y"], 1e-12, "float64 product == float64 reference (out)")     # ... IS the reference
    assert_close(gx64, f64["d32/gx"], 1e-12, "float64 product == float64 reference (gx)")
    dg = d.to(DEV)
    x = cu(g["x"]).requires_grad_(True)
    y, outs = forward_with_activations(dg, x)
    (gx,) = torch.autograd.grad(torch.nn.functional.softplus(-y).mean(), x)
    assert_close(y, f64["d32/y"], _f64_bar(f64, "d32/fp32_err_y"), "D out vs float64")
    flips = gate_flips(outs, outs64)
    assert len(flips) <= 4, f"{len(flips)} LeakyReLU gates disagree with float64"
    assert all(rel < 1e-5 for _, _, rel in flips), f"gate disagreement above rounding level: {flips}"
    excl = reach_mask(outs64, flips, x64) if flips else torch.zeros(4, 1, 32, 32, dtype=torch.bool)
    assert int(excl.sum()) <= 9 * 16 * max(1, len(flips))
    err = (gx.detach().double().cpu() - f64["d32/gx"]).abs() / f64["d32/gx"].abs().max()
    bar = _f64_bar(f64, "d32/fp32_err_gx")
    assert float(err.masked_fill(excl, 0).max()) <= bar, f"D input grad outside flipped-gate fields: {float(err.masked_fill(excl, 0).max()):.3e}"
    assert float(err.max()) <= 5e-3, f"D input grad inside a flipped gate's field: {float(err.max()):.3e}"


def test_tiny_generator_and_kd_step_vs_float64_reference():
    f64 = load_npz("float64_refs")
    g = load_npz("generator_tiny")
    cfg = load_json("generator_tiny_keys")["config"]
    gen = M.Generator(cfg["size"], cfg["style_dim"], cfg["n_mlp"], generator_net_shape=cfg["shape"])
    gen.load_state_dict(sub(g, "sd/"), strict=True)
    gen = gen.to(DEV)
    img = gen([cu(g["z0"])], randomize_noise=False)
    img.abs().mean().backward()
    assert_close(img, f64["gen/img"], _f64_bar(f64, "gen/fp32_err_img"), "tiny G image vs float64")
    for n, p in gen.named_parameters():
        assert_close(p.grad, f64["gen/grad/" + n], _f64_bar(f64, "gen/fp32_err/" + n), "tiny G grad vs float64 " + n)
    # KD step 0 (the reference's own G_Loss_BackProp replayed in float64)
    kg = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(kg, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(kg, "teacher_sd/"), strict=True)
    disc = M.Discriminator(32)
    disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), kg["d_seed"]), strict=True)
    student, teacher, disc = student.to(DEV), teacher.to(DEV), disc.to(DEV)
    step = kd.KDStep(student, teacher, disc, latent=24)
    st = meta["steps"][0]
    nl = student.num_layers
    g_loss, kd_l1, _ = step.g_losses([cu(kg[f"step0/z{i}"]) for i in range(st["n_z"])], st["inject_index"], cu(kg["mask"]),
                                     [cu(kg[f"step0/student_noise{i}"]) for i in range(nl)],
                                     [cu(kg[f"step0/teacher_noise{i}"]) for i in range(nl)])
    (g_loss + kd_l1).backward()
    assert abs(g_loss.item() - float(f64["kd/g_loss"])) < 1e-4 and abs(kd_l1.item() - float(f64["kd/kd_l1_loss"])) < 1e-4
    for n, p in student.named_parameters():
        assert_close(p.grad, f64["kd/grad/" + n], _f64_bar(f64, "kd/fp32_err/" + n), "KD grad vs float64 " + n)


def test_kd_step_golden_losses_grads_adam():
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(g, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
    disc = M.Discriminator(32)
    disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
    student, teacher, disc = student.to(DEV), teacher.to(DEV), disc.to(DEV)
    step = kd.KDStep(student, teacher, disc, latent=24)
    for st in meta["steps"]:
        p = f"step{st['step']}/"
        nl = student.num_layers
        zs = [cu(g[p + f"z{i}"]) for i in range(st["n_z"])]
        losses = step.g_step(zs, st["inject_index"], cu(g["mask"]),
                             student_noise=[cu(g[p + f"student_noise{i}"]) for i in range(nl)],
                             teacher_noise=[cu(g[p + f"teacher_noise{i}"]) for i in range(nl)])
        assert abs(losses["g"].item() - float(g[p + "g_loss"])) < 1e-3 * max(1.0, abs(float(g[p + "g_loss"])))
        assert abs(losses["kd_l1_loss"].item() - float(g[p + "kd_l1_loss"])) < 1e-3 * max(1.0, abs(float(g[p + "kd_l1_loss"])))
        params = dict(student.named_parameters())
        for k, v in sub(g, p + "grad/").items():
            assert_close(params[k].grad, v, TOL if v.numel() > 1 else 5e-3, f"step{st['step']} grad {k}")
        with torch.no_grad():
            for k, v in sub(g, p + "param_after/").items():
                params[k].copy_(cu(v))


def test_pruned_256_generator_vs_oracle_image_and_every_grad():
    """The parity gate of SURVEY.md §8-d at the real student shape: fixed latents, fixed noise, bs 2."""
    torch.manual_seed(5)
    c = load_json("contract_256")
    net = M.Generator(256, 512, 8, generator_net_shape=c["pruned_shape"])
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    z = [torch.randn(2, 512), torch.randn(2, 512)]
    names = [n for n, _ in net.named_parameters()]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sdr = dict(sd)
    sdr.update(leaves)
    img_r = ref_model.generator_forward_ref(sdr, z, inject_index=5, randomize_noise=False)
    gr = torch.autograd.grad(img_r.abs().mean(), [leaves[k] for k in names], allow_unused=True)
    netg = net.to(DEV)
    img_g = netg([cu(z[0]), cu(z[1])], inject_index=5, randomize_noise=False)
    assert_close(img_g, img_r, TOL, "image")
    img_g.abs().mean().backward()
    params = dict(netg.named_parameters())
    for k, b in zip(names, gr):
        a = params[k].grad
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        assert_close(a, b, TOL if b.numel() > 1 else 5e-3, "grad " + k)


def test_pruned_1024_generator_vs_oracle_image_and_grads():
    """BASELINE configs[3] shape: 1024 px 70 %-pruned student [154x10,77,77,39,39,20,20,10,10], bs 1: image and every
    parameter gradient.  At this size (10^7 LeakyReLU gates on a random-init net) fp32 itself is only reproducible to
    ~1e-3 on the gradients: the fp32 CPU oracle differs from its own float64 evaluation by up to 2e-3.  So the truth
    here is the oracle in float64; the image must meet the 1e-3 bar, the gradients 5e-3 (the fp32 CPU reference's own
    spread against float64 reaches 2.4e-3 on this net; HIP path observed: 3e-4 on most conv layers, up to 1.5e-3 on the
    mapping-network gradients that sum every layer's contribution).  That this is gate-flip chaos and not kernel error
    is pinned separately: test_styled_conv_layers_vs_float64 holds every layer to 5e-6 of float64."""
    torch.manual_seed(11)
    shape = [154] * 10 + [77, 77, 39, 39, 20, 20, 10, 10]
    net = M.Generator(1024, 512, 8, generator_net_shape=shape)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    z = [torch.randn(1, 512), torch.randn(1, 512)]
    proj = torch.randn(1, 3, 1024, 1024)   # linear functional of the image
    names = [n for n, _ in net.named_parameters()]

    def ref(dtype):
        leaves = {k: sd[k].to(dtype).clone().requires_grad_(True) for k in names}
        sdr = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        sdr.update(leaves)
        img = ref_model.generator_forward_ref(sdr, [a.to(dtype) for a in z], inject_index=7, randomize_noise=False)
        return img.detach(), torch.autograd.grad((img * proj.to(dtype)).mean(), [leaves[k] for k in names], allow_unused=True)

    img32, g32 = ref(torch.float32)
    with ref_ops.gates() as rec:
        img64, g64 = ref(torch.float64)
    assert tuple(img64.shape) == (1, 3, 1024, 1024)
    netg = net.to(DEV)
    # the HIP run's LeakyReLU gates in the oracle's call order: mapping network of z0, of z1 (the product maps the stacked
    # latents in one pass: rows 0 / 1 of each layer's output), then the 17 styled convs
    acts_map, acts_conv, hooks = [], [], []
    for m in netg.style:
        if isinstance(m, M.EqualLinear):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: acts_map.append(out.detach())))
    for m in [netg.conv1] + list(netg.convs):
        hooks.append(m.register_forward_hook(lambda mod, inp, out: acts_conv.append((out[0] if isinstance(out, tuple) else out).detach())))
    img_g = netg([cu(z[0]), cu(z[1])], inject_index=7, randomize_noise=False)
    for h in hooks:
        h.remove()
    gates_g = [(a[:1] > 0).cpu() for a in acts_map] + [(a[1:] > 0).cpu() for a in acts_map] + [(a > 0).cpu() for a in acts_conv]
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-300))
    assert rel(img_g.detach(), img64) <= TOL, "image 1024"
    (img_g * cu(proj)).mean().backward()
    params = dict(netg.named_parameters())
    for k, b32, b64 in zip(names, g32, g64):
        a = params[k].grad
        if b64 is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        e_gpu, e_cpu = rel(a, b64), rel(b32, b64)
        bar = max(5e-3, 3 * e_cpu) if b64.numel() > 1 else max(5e-2, 5 * e_cpu)   # noise.weight: one cancelling sum over up to 10^6 pixels
        assert e_gpu <= bar, f"grad {k}: HIP vs float64 {e_gpu:.2e}, fp32 CPU reference vs float64 {e_cpu:.2e}"
    # ... and the north-star bar proper, on the common gate pattern: every gate where the HIP run and float64
    # disagree sits at rounding level of its layer; float64 evaluated on the HIP run's pattern then agrees with every
    # gradient to 5e-5 (observed 6.5e-6; single-element cancelling sums 1e-3) — what the 5e-3 above leaves room for is gate flips only
    n_dis = ref_ops.gate_disagreements(rec, gates_g, rounding=1e-5, max_fraction=1e-5)
    with ref_ops.gates(force=gates_g):
        _, g64f = ref(torch.float64)
    worst = 0.0
    for k, b in zip(names, g64f):
        if b is None:
            continue
        e = rel(params[k].grad, b)
        worst = max(worst, e if b.numel() > 1 else 0.0)
        assert e <= (5e-5 if b.numel() > 1 else 1e-3), f"grad {k} on the common gate pattern ({n_dis} disagreements): {e:.2e}"
    print(f"1024 px student: {n_dis} gate disagreements with float64, worst tensor gradient error on the common pattern {worst:.2e}")


@pytest.mark.parametrize("cfg", [(154, 154, 4, True, 16), (154, 154, 8, False, 1), (154, 154, 16, True, 1),
                                 (512, 512, 4, True, 2), (154, 154, 32, True, 1), (77, 39, 32, False, 2),
                                 # the real teacher / student layer shapes of configs[1] (every kernel plan the bench step selects:
                                 # register-direct transposed conv, Winograd wide / 4-wave, register-direct weight gradient)
                                 (512, 512, 32, False, 2), (512, 512, 64, False, 1), (512, 256, 64, True, 1), (256, 256, 128, False, 1),
                                 (256, 128, 128, True, 1), (128, 128, 256, False, 1), (154, 77, 64, True, 2), (77, 77, 128, False, 1),
                                 (77, 39, 128, True, 1), (39, 39, 256, False, 1)])
def test_styled_conv_layers_vs_float64(cfg, wino4_policy):
    """Single StyledConv layers (direct, transposed and Winograd paths) against the oracle evaluated in float64:
    output and every gradient within 5e-6 — the fp32 MFMA path is as accurate as the fp32 CPU reference (2-7e-7)."""
    cin, cout, H, up, B = cfg
    torch.manual_seed(3)
    m = M.StyledConv(cin, cout, 3, 512, upsample=up)
    with torch.no_grad():
        m.noise.weight.fill_(0.1)
        m.activate.bias.copy_(0.1 * torch.randn(cout))
    sd = {"l." + k: v.detach().double() for k, v in m.state_dict().items() if v.is_floating_point()}
    Ho = 2 * H if up else H
    x, w = torch.randn(B, cin, H, H), torch.randn(B, 512)
    noise, go = torch.randn(B, 1, Ho, Ho), torch.randn(B, cout, Ho, Ho)
    names = [n for n, _ in m.named_parameters()]
    leaves = {"l." + k: sd["l." + k].clone().requires_grad_(True) for k in names}
    sdr = dict(sd)
    sdr.update(leaves)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    mg = m.to(DEV)
    xg, wg = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    yg = mg(xg, wg, noise=cu(noise))
    gg = torch.autograd.grad(yg, [xg, wg] + [dict(mg.named_parameters())[k] for k in names], cu(go))
    # common-gate protocol (oracle/ref_ops.py `gates`): at the real layer sizes (10^6 .. 10^7 LeakyReLU gates) a few
    # pre-activations sit within fp32 rounding of 0; each such gate must be at rounding level, and the float64 oracle is
    # evaluated on the HIP run's gate pattern — the same piecewise-linear function — where everything agrees to 5e-6
    gpu_gate = [(yg.detach() > 0).cpu()]
    with ref_ops.gates() as rec:
        with torch.no_grad():
            ref_model._styled_conv(sd, "l", x.double(), w.double(), noise.double(), upsample=up)
    f4 = wino_f4(cin, cout, H, H, up)
    bar = 5e-5 if f4 else 5e-6
    ref_ops.gate_disagreements(rec, gpu_gate, rounding=1e-4 if f4 else 1e-5, max_fraction=1e-4 if f4 else 1e-5)
    with ref_ops.gates(force=gpu_gate):
        yr, _ = ref_model._styled_conv(sdr, "l", xr, wr, noise.double(), upsample=up)
    gr = torch.autograd.grad(yr, [xr, wr] + [leaves["l." + k] for k in names], go.double())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-300))
    assert rel(yg.detach(), yr.detach()) <= bar, f"{cfg} out {rel(yg.detach(), yr.detach()):.2e}"
    for nm, a, b in zip(["x", "style"] + names, gg, gr):
        # noise.weight: ONE number, the sum of up to 10^7 signed products (fp32 accumulation of a cancelling sum)
        assert rel(a, b) <= (bar if b.numel() > 1 else 1e-3), f"{cfg} grad {nm}: {rel(a, b):.2e}"


def test_full_256_teacher_forward_vs_oracle(wino4_policy):
    """Full 256 px teacher, all seven RGB outputs vs the oracle — under the per-launch Winograd policy (B = 1: the 32^2 / 64^2
    layers on F(2x2)) and with F(4x4) forced on every eligible layer (the kernels the bs-16 bench's launches select)."""
    torch.manual_seed(6)
    net = M.Generator(256, 512, 8)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    z = [torch.randn(1, 512)]
    with torch.no_grad():
        img_r = ref_model.generator_forward_ref(sd, z, randomize_noise=False, return_rgb_list=True)
        img_g = net.to(DEV)([cu(z[0])], randomize_noise=False, return_rgb_list=True)
    for i, (a, b) in enumerate(zip(img_g, img_r)):
        assert_close(a, b, TOL, f"teacher rgb {i}")


def test_size_independent_properties_at_full_size():
    """bs 16 student shapes (too slow for the CPU oracle): linearity of the conv in x, batch independence."""
    torch.manual_seed(7)
    m = M.StyledConv(39, 39, 3, 512).to(DEV)
    with torch.no_grad():
        m.noise.weight.fill_(0.0)
        m.activate.bias.zero_()
    x = torch.randn(16, 39, 256, 256, device=DEV)
    w = torch.randn(16, 512, device=DEV)
    with torch.no_grad():
        y = m(x, w)
        # LeakyReLU is positively homogeneous and bias/noise are zero: f(2x) == 2 f(x)
        assert_close(m(2 * x, w), 2 * y, 1e-5, "homogeneity")
        # sample 3 alone == sample 3 within the batch
        assert_close(m(x[3:4], w[3:4]), y[3:4], 1e-6, "batch independence")


@pytest.mark.parametrize("deterministic", [False, True])
def test_graphed_kd_step_matches_eager_step(deterministic):
    """(deterministic=True: `cagc_set_tuning("deterministic", 1)` before warm-up and capture — the replayed step must then
    reproduce the eager one BIT for bit: no fp32-atomic K split, every backward reduction through the order-independent sink
    whose scratch the warm-up steps size before the capture starts.)

    HIP-graph replay of the step (cagc.kd.GraphedKDStep) == the eager step on the same inputs, two steps in a
    row (the second checks that Adam state and the device-side mixing index advance correctly under replay)."""
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")

    def build():
        student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
        student.load_state_dict(sub(g, "student_sd/"), strict=True)
        teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
        teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
        disc = M.Discriminator(32)
        disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
        return student.to(DEV), teacher.to(DEV), disc.to(DEV)

    se, te, de = build()
    sg, tg, dg = build()
    eager = kd.KDStep(se, te, de, latent=24)
    B = g["mask"].shape[0]
    prev = _lib.set_tuning("deterministic", 1) if deterministic else _lib.get_tuning("deterministic")
    deterministic = deterministic or bool(prev)       # the whole suite may run under CAGC_DETERMINISTIC=1
    try:
        graphed = kd.GraphedKDStep(sg, tg, dg, B, cu(g["mask"]), random_noise=False, latent=24)
        _graph_vs_eager_steps(g, meta, se, sg, eager, graphed, deterministic)
    finally:
        _lib.set_tuning("deterministic", prev)


def _graph_vs_eager_steps(g, meta, se, sg, eager, graphed, deterministic):
    # capture (incl. its warm-up steps) must leave the student and the Adam state untouched
    for k, v in sub(g, "student_sd/").items():
        assert torch.equal(dict(sg.state_dict())[k].cpu(), v), f"capture changed {k}"
    for st_ in graphed.optim.state.values():
        for v in st_.values():
            if torch.is_tensor(v):
                assert float(v.abs().sum()) == 0.0
    nl = se.num_layers
    for st in meta["steps"]:
        p = f"step{st['step']}/"
        zs = [cu(g[p + f"z{i}"]) for i in range(st["n_z"])]
        sn = [cu(g[p + f"student_noise{i}"]) for i in range(nl)]
        tn = [cu(g[p + f"teacher_noise{i}"]) for i in range(nl)]
        le = eager.g_step(zs, st["inject_index"], cu(g["mask"]), sn, tn)
        lg = graphed.g_step(zs, st["inject_index"], cu(g["mask"]), sn, tn)
        torch.cuda.synchronize()
        assert abs(le["g"].item() - lg["g"].item()) < 1e-5 and abs(le["kd_l1_loss"].item() - lg["kd_l1_loss"].item()) < 1e-5
        pe, pg = dict(se.named_parameters()), dict(sg.named_parameters())
        for k in pe:
            first = st is meta["steps"][0]   # later steps start from mapping-network weights that differ in the last bits
            if deterministic and first and not k.startswith("style."):
                assert torch.equal(pg[k].grad, pe[k].grad), f"deterministic mode: graph grad {k} differs by {float((pg[k].grad - pe[k].grad).abs().max()):.3e}"
                assert torch.equal(pg[k].detach(), pe[k].detach()), f"deterministic mode: graph param {k}"
                continue
            if deterministic and first:   # mapping network = library GEMMs (torch.addmm -> rocBLAS / hipBLASLt), whose kernel choice under
                # stream capture is the library's: its weight gradients agree to the last bits, not bit for bit
                assert_close(pg[k].grad, pe[k].grad, 2e-6, f"graph grad {k}")
                assert_close(pg[k].detach(), pe[k].detach(), 2e-6, f"graph param {k}")
                continue
            assert_close(pg[k].grad, pe[k].grad, 5e-4 if pe[k].numel() > 1 else 3e-3, f"graph grad {k}")  # atomics order differs run to run
            assert_close(pg[k].detach(), pe[k].detach(), 1e-4, f"graph param {k}")


@pytest.mark.parametrize("cfg", [(2, 128, 256, 64, 64), (3, 20, 36, 18, 18), (1, 512, 512, 8, 8), (16, 128, 256, 256, 256)])
def test_discriminator_downsample_conv_vs_oracle(cfg):
    """Blur(pad 2,2) -> 3x3 stride-2 EqualConv2d -> FusedLeakyReLU on the hand-written MFMA path vs the oracle
    (forward + input gradient + weight gradient through the stock fallback)."""
    B, cin, cout, H, W = cfg
    torch.manual_seed(8)
    layer = M.ConvLayer(cin, cout, 3, downsample=True)
    with torch.no_grad():
        layer[2].bias.copy_(0.1 * torch.randn(cout))
    sd = {"l." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    big = B * cin * H * W > 50_000_000
    lg = layer.to(DEV)
    xg = cu(x).requires_grad_(True)
    yg = lg(xg)
    go = torch.randn(yg.shape)
    if big:   # CPU oracle too slow at bs 16 x 256^2: linearity of the (conv o blur) in x instead, on the raw conv
        for p in lg.parameters():
            p.requires_grad_(False)
        with torch.no_grad():
            lg[2].bias.zero_()
            y1 = lg(cu(x))
            y2 = lg(2 * cu(x))
        assert_close(y2, 2 * y1, 1e-5, "homogeneity")
        return
    xr = x.clone().requires_grad_(True)
    yr = ref_model._conv_layer(sd, "l", xr, 3, downsample=True)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    assert_close(yg, yr, TOL, "out")
    gxg, gwg = torch.autograd.grad(yg, [xg, lg[1].weight], cu(go))
    assert_close(gxg, gxr, TOL, "grad x")
    wr = sd["l.1.weight"].clone().requires_grad_(True)
    sd2 = dict(sd)
    sd2["l.1.weight"] = wr
    (gwr,) = torch.autograd.grad(ref_model._conv_layer(sd2, "l", x, 3, downsample=True), wr, go)
    assert_close(gwg, gwr, TOL, "grad weight")


@pytest.mark.parametrize("cfg", [(2, 128, 128, 64, 64), (3, 20, 36, 32, 32), (1, 512, 512, 32, 64)])
def test_discriminator_conv3x3_act_winograd_vs_oracle(cfg):
    """EqualConv2d(3x3, pad 1) -> FusedLeakyReLU on the Winograd F(2x2,3x3) MFMA kernel (forward, input gradient via the
    Winograd data gradient, bias gradient, weight gradient through the stock fallback) vs the oracle."""
    B, cin, cout, H, W = cfg
    torch.manual_seed(9)
    layer = M.ConvLayer(cin, cout, 3)
    with torch.no_grad():
        layer[1].bias.copy_(0.1 * torch.randn(cout))
    sd = {"l." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in ("l.0.weight", "l.1.bias")}
    sdr = dict(sd)
    sdr.update(leaves)
    xr = x.clone().requires_grad_(True)
    yr = ref_model._conv_layer(sdr, "l", xr, 3)
    go = torch.randn(yr.shape)
    gr = torch.autograd.grad(yr, [xr, leaves["l.0.weight"], leaves["l.1.bias"]], go)
    lg = layer.to(DEV)
    xg = cu(x).requires_grad_(True)
    yg = lg(xg)
    assert_close(yg, yr, TOL, "out")
    gg = torch.autograd.grad(yg, [xg, lg[0].weight, lg[1].bias], cu(go))
    for nm, a, b in zip(("x", "weight", "bias"), gg, gr):
        assert_close(a, b, TOL, f"{cfg} grad {nm}")


@pytest.mark.parametrize("cfg", [(2, 128, 128, 64, 64), (3, 20, 36, 32, 32), (2, 512, 512, 32, 32), (16, 128, 128, 256, 256)])
def test_frozen_conv3x3_act_data_gradient_fused_into_winograd_staging(cfg):
    """Frozen D layer (generator step): cagc_wino_conv3x3_act_dgrad (LeakyReLU backward applied while the Winograd kernel
    stages its input) == the two-pass path (cagc_fused_bias_act_bwd, then the Winograd data gradient), BIT for bit — the
    staged product gout * lrelu'(out) is the same fp32 multiplication."""
    from cagc.op import modconv as mc
    B, cin, cout, H, W = cfg
    torch.manual_seed(10)
    layer = M.ConvLayer(cin, cout, 3).to(DEV)
    with torch.no_grad():
        layer[1].bias.copy_(0.1 * torch.randn(cout))
    kd.requires_grad(layer, False)
    x = torch.randn(B, cin, H, W, device=DEV, requires_grad=True)
    go = torch.randn(B, cout, H, W, device=DEV)
    outs = {}
    for fused in (True, False):
        mc.FUSE_ACT_DGRAD = fused
        try:
            (outs[fused],) = torch.autograd.grad(layer(x), x, go)
        finally:
            mc.FUSE_ACT_DGRAD = True
    assert torch.equal(outs[True], outs[False])
    assert torch.isfinite(outs[True]).all() and float(outs[True].abs().max()) > 0


@pytest.mark.parametrize("cfg", [(2, 128, 256, 64, 64), (3, 20, 36, 18, 22), (1, 512, 512, 8, 8), (4, 128, 256, 128, 128)])
def test_discriminator_skip_blur_down_conv1x1_vs_oracle(cfg):
    """ResBlock skip: Blur(pad 1,1) -> 1x1 stride-2 EqualConv2d (no bias / activation) on the fused path (decimating FIR +
    1x1 MFMA GEMM) vs the oracle: forward, input gradient, weight gradient."""
    B, cin, cout, H, W = cfg
    torch.manual_seed(9)
    layer = M.ConvLayer(cin, cout, 1, downsample=True, activate=False, bias=False)
    assert layer._fused_skip
    sd = {"l." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    lg = layer.to(DEV)
    xg = cu(x).requires_grad_(True)
    yg = lg(xg)
    assert yg.grad_fn is not None and "BlurDownConv1x1" in type(yg.grad_fn).__name__
    go = torch.randn(yg.shape)
    xr = x.clone().requires_grad_(True)
    wr = sd["l.1.weight"].clone().requires_grad_(True)
    sd["l.1.weight"] = wr
    yr = ref_model._conv_layer(sd, "l", xr, 1, downsample=True, activate=False, bias=False)
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], go)
    assert_close(yg, yr, TOL, "out")
    gxg, gwg = torch.autograd.grad(yg, [xg, lg[1].weight], cu(go))
    assert_close(gxg, gxr, TOL, "grad x")
    assert_close(gwg, gwr, TOL, "grad weight")


def test_c_abi_error_contract():
    """errors come back as negative codes + cagc_last_error(), never as exceptions across the ABI; the
    Python binding turns them into RuntimeError.  Also: edge shapes the reference accepts (empty batch planes)."""
    x = torch.randn(1, 8, 10, 12, device=DEV)
    out = torch.empty(1, 16, 10, 12, device=DEV)
    up = torch.empty(_lib.query("cagc_wino_packed_elems", 8, 16), device=DEV)
    with pytest.raises(RuntimeError, match="H % 8 == 0"):      # ineligible size for the Winograd entry point
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, 1, 8, 16, 10, 12, 0, None, None, 0, None,
                  None, 0.2, 1.0)
    with pytest.raises(RuntimeError, match="null tensor"):
        _lib.call("cagc_upfirdn2d", None, _lib.ptr(x), _lib.ptr(x), 8, 10, 12, 10, 12, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="expected"):         # wrong output size
        k = torch.ones(4, 4, device=DEV) / 16
        _lib.call("cagc_upfirdn2d", _lib.ptr(out), _lib.ptr(x), _lib.ptr(k), 8, 10, 12, 10, 12, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1)
    lib = _lib.load()
    assert lib.cagc_last_error().decode() != ""
    # zero planes: a no-op that succeeds (the reference's ops accept empty batches)
    k = torch.ones(4, 4, device=DEV) / 16
    _lib.call("cagc_upfirdn2d", _lib.ptr(out), _lib.ptr(x), _lib.ptr(k), 0, 10, 12, 9, 11, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1)
    e = torch.empty(0, 8, 10, 12, device=DEV)
    assert fused_leaky_relu(e, torch.zeros(8, device=DEV)).shape == e.shape
    assert upfirdn2d(e, k, pad=(1, 1)).shape == (0, 8, 9, 11)


# ---------------------------------------------------------------------------------------------------
# GraphedKDStep with world_size = 2: two processes on this one GPU (CAGC_SINGLE_DEVICE=1, gloo), each capturing its own
# HIP graphs and joining the flat-gradient all-reduce between them == one process on the concatenated batch.
# ---------------------------------------------------------------------------------------------------
def _graph2_inputs(g, meta):
    import torch as _t
    gen = _t.Generator().manual_seed(5)
    nl = 7
    return dict(z=[_t.randn(8, 24, generator=gen), _t.randn(8, 24, generator=gen)], mask=g["mask"].repeat(2, 1, 1, 1),
                sn=[_t.randn(8, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(nl)],
                tn=[_t.randn(8, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(nl)])


def _graph2_models(g, meta):
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(g, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
    disc = M.Discriminator(32)
    disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
    return student.to(DEV), teacher.to(DEV), disc.to(DEV)


def _graph2_worker(rank, world, port, tmp):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CAGC_SINGLE_DEVICE="1", CAGC_DIST_BACKEND="gloo")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "content-aware-gan-compression_amd"), os.path.dirname(os.path.abspath(__file__))):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from cagc import distributed as cd
    cd.init_from_env()
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _graph2_models(g, meta)
    d = _graph2_inputs(g, meta)
    sl = slice(rank, None, world)          # rank r takes samples r::2: keeps D's minibatch-stddev groups identical
    # this rank's LOCAL gradient of the un-scaled loss, eagerly, on a copy of the student: the captured step's reduced flat gradient
    # must be the MEAN of these over the ranks (the 1 / world_size lives only in the backward seed — nothing may scale twice)
    import copy
    probe = kd.KDStep(copy.deepcopy(student), teacher, disc, latent=24)
    args = ([cu(z[sl]) for z in d["z"]], 3, cu(d["mask"][sl]), [cu(n[sl]) for n in d["sn"]], [cu(n[sl]) for n in d["tn"]])
    kd.requires_grad(probe.student, True)
    kd.requires_grad(disc, False)
    total, _, _ = probe.g_total(*args)
    total.backward()
    local = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone() for k, p in probe.student.named_parameters()}
    mean = {}
    for k, v in local.items():
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        mean[k] = sum(parts) / world
    step = kd.GraphedKDStep(student, teacher, disc, 4, cu(d["mask"][sl]), random_noise=False, world_size=world, latent=24)
    assert step.comm == "host" and "nccl" in step.comm_reason      # gloo: collectives are not capturable; the reason is recorded
    for it in range(2):
        step.g_step(*args)
        if it == 0:
            torch.cuda.synchronize()
            names = {id(p): k for k, p in student.named_parameters()}
            for p_, view in zip(step._params, step._grad_views):
                k = names[id(p_)]
                assert_close(view, mean[k], 5e-4 if view.numel() > 1 else 3e-3, f"rank {rank}: reduced flat gradient vs mean of local gradients: {k}")
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({k: v.detach().cpu() for k, v in student.named_parameters()}, tmp)
    cd.barrier()
    dist.destroy_process_group()


def test_graphed_kd_step_two_processes_equal_one_process(tmp_path):
    import os
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "graph2.pt")
    mp.spawn(_graph2_worker, args=(2, 29500 + (os.getpid() % 90), tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _graph2_models(g, meta)
    d = _graph2_inputs(g, meta)
    step = kd.KDStep(student, teacher, disc, latent=24)
    for _ in range(2):
        step.g_step([cu(z) for z in d["z"]], 3, cu(d["mask"]), [cu(n) for n in d["sn"]], [cu(n) for n in d["tn"]])
    for k, p in student.named_parameters():
        # two Adam steps (beta1 = 0: update = lr * g / sqrt(v)) on gradients that differ at atomics / summation-order level
        assert_close(got[k], p.detach(), 2e-3, "2-process graph replay param " + k)


@pytest.mark.parametrize("cfg", [(2, 128, 256, 64, 64), (3, 64, 64, 32, 32), (16, 128, 256, 256, 256)])
def test_frozen_resblock_single_node_matches_layerwise_path(cfg):
    """Frozen D ResBlock as ONE autograd node (merge scale folded into the activation backward / adjoint FIR, skip-branch
    gradient added inside the Winograd data-gradient kernel's store) vs the layer-by-layer path: identical forward, input
    gradient equal up to the re-association of the 1/sqrt(2) factor."""
    from cagc.op import modconv as mc
    B, cin, cout, H, W = cfg
    torch.manual_seed(13)
    blk = M.ResBlock(cin, cout).to(DEV)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn_like(p))
    kd.requires_grad(blk, False)
    x = torch.randn(B, cin, H, W, device=DEV, requires_grad=True)
    go = torch.randn(B, cout, H // 2, W // 2, device=DEV)
    res = {}
    for fused in (True, False):
        mc.FUSE_RESBLOCK = fused
        try:
            assert blk._frozen_fast_path(x) == fused
            y = blk(x)
            (gx,) = torch.autograd.grad(y, x, go)
            res[fused] = (y.detach(), gx)
        finally:
            mc.FUSE_RESBLOCK = True
    assert_close(res[True][0], res[False][0], 1e-6, f"{cfg} frozen ResBlock output")   # same kernels; split-K atomics at small sizes
    assert_close(res[True][1], res[False][1], 5e-5, f"{cfg} frozen ResBlock input gradient")   # F(4x4) conv1 on both paths, different summation order


@pytest.mark.parametrize("cfg", [(2, 128, 64, 64), (3, 36, 20, 24), (16, 128, 256, 256)])
def test_from_rgb_streaming_kernels_match_implicit_gemm_path(cfg):
    """Frozen from-RGB layer (cagc_fromrgb_fwd / cagc_fromrgb_act_dgrad) vs the implicit-GEMM path with the fused-act
    backward (trainable flags on), and vs float64."""
    B, C, H, W = cfg
    torch.manual_seed(14)
    layer = M.ConvLayer(3, C, 1)
    with torch.no_grad():
        layer[1].bias.copy_(0.1 * torch.randn(C))
    x = torch.randn(B, 3, H, W)
    go = torch.randn(B, C, H, W)
    xr = x.double().requires_grad_(True)
    yr = ref_ops.fused_leaky_relu_ref(torch.nn.functional.conv2d(xr, layer[0].weight.detach().double() * layer[0].scale),
                                      layer[1].bias.detach().double())
    (gr,) = torch.autograd.grad(yr, xr, go.double())
    lg = layer.to(DEV)
    res = {}
    for frozen in (True, False):
        kd.requires_grad(lg, not frozen)
        xg = cu(x).requires_grad_(True)
        yg = lg(xg)
        (gg,) = torch.autograd.grad(yg, xg, cu(go))
        res[frozen] = (yg.detach(), gg)
        assert_close(yg, yr, 2e-6, f"{cfg} from-RGB out (frozen={frozen})")
        # a 1x1 layer's input gradient at a pixel depends only on that pixel's LeakyReLU gates: pixels where fp32 rounding put
        # a gate on the other side of 0 than float64 (pre-activation ~1e-7, a dozen of 134M elements at the largest size)
        # are excluded, and must be rare and at rounding level
        flip = ((yg.detach().cpu() > 0) != (yr.detach() > 0))
        assert int(flip.sum()) <= max(4, 1e-6 * flip.numel())
        if flip.any():
            assert float(yr.detach()[flip].abs().max()) < 1e-5 * float(yr.detach().abs().max())
        keep = ~flip.any(1, keepdim=True)
        err = ((gg.double().cpu() - gr).abs() * keep).max().item() / gr.abs().max().item()
        assert err <= 2e-5, f"{cfg} from-RGB input gradient (frozen={frozen}): {err:.3e}"


@pytest.mark.parametrize("up", [False, True])
def test_styled_conv_noise_gradient(up):
    """A caller that optimises the per-layer noise maps (a projector) gets d out / d noise from the fused op: per-sample
    [B,1,H,W] and shared [1,1,H,W] maps, vs the composed CPU formulation."""
    torch.manual_seed(15)
    m = M.StyledConv(20, 12, 3, 32, upsample=up)
    with torch.no_grad():
        m.noise.weight.fill_(0.3)
        m.activate.bias.copy_(0.1 * torch.randn(12))
    B, H = 3, 16
    oh = 2 * H if up else H
    x, w = torch.randn(B, 20, H, H), torch.randn(B, 32)
    go = torch.randn(B, 12, oh, oh)
    for nb in (B, 1):
        noise = torch.randn(nb, 1, oh, oh)
        nr = noise.clone().requires_grad_(True)
        (gref,) = torch.autograd.grad(m(x, w, noise=nr), nr, go)
        mg = M.StyledConv(20, 12, 3, 32, upsample=up)
        mg.load_state_dict(m.state_dict())
        mg = mg.to(DEV)
        ng = cu(noise).requires_grad_(True)
        (gg,) = torch.autograd.grad(mg(cu(x), cu(w), noise=ng), ng, cu(go))
        assert tuple(gg.shape) == tuple(noise.shape)
        assert_close(gg, gref, 2e-5, f"noise gradient (up={up}, noise batch {nb})")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float16, 2e-3)])
def test_ops_fp64_fp16_dispatch(dtype, tol):
    """The two reference operators dispatch over fp32 / fp64 / fp16 (op/fused_bias_act_kernel.cu:79, op/upfirdn2d_kernel.cu:311):
    forward, backward and double backward of both on the any-dtype kernels vs the CPU formulation in float64."""
    torch.manual_seed(16)
    x = torch.randn(3, 7, 10, 9, dtype=torch.float64)
    b = torch.randn(7, dtype=torch.float64)
    go, ggi = torch.randn_like(x), torch.randn_like(x)

    def act(x, b, go, ggi):
        x, b, go = x.requires_grad_(True), b.requires_grad_(True), go.requires_grad_(True)
        y = fused_leaky_relu(x, b)
        gx, gb = torch.autograd.grad(y, [x, b], go, create_graph=True)
        (ggo,) = torch.autograd.grad(gx, go, ggi)
        return [y, gx, gb, ggo]

    ref = act(x.clone(), b.clone(), go.clone(), ggi.clone())
    got = act(cu(x).to(dtype), cu(b).to(dtype), cu(go).to(dtype), cu(ggi).to(dtype))
    for nm, a, r in zip(("y", "gx", "gbias", "ggo"), got, ref):
        assert a.dtype == dtype
        assert_close(a, r, tol if nm != "gbias" else 10 * tol, f"fused_leaky_relu {dtype} {nm}")
    k = M.make_kernel([1, 3, 3, 1]).double()
    for up, down, pad in ((1, 1, (2, 2)), (2, 1, (2, 1)), (1, 2, (1, 1))):
        xr = x.clone().requires_grad_(True)
        yr = upfirdn2d(xr, k, up=up, down=down, pad=pad)
        g2 = torch.randn_like(yr)
        (gr,) = torch.autograd.grad(yr, xr, g2)
        xg = cu(x).to(dtype).requires_grad_(True)
        yg = upfirdn2d(xg, cu(k).to(dtype), up=up, down=down, pad=pad)
        (gg,) = torch.autograd.grad(yg, xg, cu(g2).to(dtype))
        assert yg.dtype == dtype
        assert_close(yg, yr, tol, f"upfirdn2d {dtype} up{up} down{down}")
        assert_close(gg, gr, tol, f"upfirdn2d {dtype} up{up} down{down} grad")


_WINO_SHAPES_SCRIPT = r"""
import os, sys, torch
import torch.nn.functional as F
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
torch.manual_seed(0)
worst = 0.0
for (B, Cin, Cout, H, W, styled) in [(2, 128, 128, 64, 64, 0), (1, 160, 200, 32, 64, 1), (1, 72, 256, 16, 64, 1), (2, 40, 384, 8, 32, 0), (3, 20, 36, 32, 32, 1), (1, 11, 7, 8, 32, 0),
                                     (2, 39, 39, 16, 96, 1)]:
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda")
    s = torch.rand(B, Cin, device="cuda") + 0.5
    d = torch.rand(B, Cout, device="cuda") + 0.5
    noise = torch.randn(B, 1, H, W, device="cuda"); nw = torch.tensor([0.3], device="cuda"); bias = torch.randn(Cout, device="cuda")
    up = mc.pack_wino(w, 1.0, False); out = torch.full((B, Cout, H, W), float("nan"), device="cuda")
    if styled:
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), _lib.ptr(s), B, Cin, Cout, H, W, 1, _lib.ptr(d), _lib.ptr(noise), B,
                  _lib.ptr(nw), _lib.ptr(bias), 0.2, 2 ** 0.5)
        ref = F.conv2d((x * s[:, :, None, None]).double(), w.double(), padding=1) * d[:, :, None, None].double()
        ref = F.leaky_relu(ref + 0.3 * noise.double() + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
    else:
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, Cin, Cout, H, W, 0, None, None, 0, None, None, 0.2, 1.0)
        ref = F.conv2d(x.double(), w.double(), padding=1)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    assert err == err and err < 5e-6, (B, Cin, Cout, H, W, styled, err)
    worst = max(worst, err)
# gated variant: data gradient of conv3x3 + bias + LeakyReLU in one launch (frozen discriminator layer), with the residual add
for (B, Cin, Cout, H, W) in [(2, 64, 128, 32, 64), (1, 24, 40, 16, 32), (1, 256, 72, 16, 64)]:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda"); gout = torch.randn(B, Cout, H, W, device="cuda")
    act = torch.randn(B, Cout, H, W, device="cuda"); res = torch.randn(B, Cin, H, W, device="cuda")
    upb = mc.pack_wino(w, 1.0, True); gx = torch.full((B, Cin, H, W), float("nan"), device="cuda")
    _lib.call("cagc_wino_conv3x3_act_dgrad", _lib.ptr(gx), _lib.ptr(gout), _lib.ptr(act), _lib.ptr(upb), _lib.ptr(res), B, Cin, Cout, H, W, 0.2, 2 ** 0.5)
    gin = gout.double() * torch.where(act > 0, 1.0, 0.2).double() * 2 ** 0.5
    ref = F.conv_transpose2d(gin, w.double(), padding=1) + res.double()
    err = ((gx.double() - ref).abs().max() / ref.abs().max()).item()
    assert err == err and err < 5e-6, ("gated", B, Cin, Cout, H, W, err)
    worst = max(worst, err)
print("WINO_OK %.3e" % worst)
"""


@pytest.mark.parametrize("nh", ["1", "2", "wide"])
def test_winograd_both_workgroup_shapes_vs_float64(nh, tmp_path):
    """The Winograd kernel picks 4-wave (NH 1), 8-wave (NH 2) or wide (128-channel, NH 3) workgroups per launch (conv_wino.hip); the choice
    is read once per process, so each shape is forced in its own interpreter: linear and styled epilogues, ragged channel
    counts (K padded to 16, M to the channel tile), against a float64 direct convolution (5e-6 of the output scale)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "wino_shapes.py"
    script.write_text(_WINO_SHAPES_SCRIPT)
    env = dict(os.environ, CAGC_WINO_NH="2", CAGC_WINO_WIDE="2") if nh == "wide" else dict(os.environ, CAGC_WINO_NH=nh, CAGC_WINO_WIDE="0")
    env["CAGC_WINO_F4"] = "0"          # this test pins the F(2x2) kernel's workgroup shapes
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WINO_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shape", [(16, 512), (3, 24), (64, 512), (1, 7)])
def test_pixelnorm_forward_and_backward_vs_float64(shape):
    """PixelNorm (reference model.py:14-24) on its own: cagc_pixelnorm_fwd / cagc_pixelnorm_bwd (one wavefront per row,
    shuffle reduction over the channels) against x * rsqrt(mean_c(x^2) + 1e-8) and its autograd gradient in float64."""
    torch.manual_seed(21)
    x = torch.randn(*shape)
    go = torch.randn(*shape)
    xr = x.double().requires_grad_(True)
    yr = xr * torch.rsqrt(torch.mean(xr * xr, dim=1, keepdim=True) + 1e-8)
    (gr,) = torch.autograd.grad(yr, xr, go.double())
    xg = cu(x).requires_grad_(True)
    yg = M.PixelNorm()(xg)
    assert yg.grad_fn is not None and "PixelNorm" in type(yg.grad_fn).__name__      # the HIP node, not the composed formula
    (gg,) = torch.autograd.grad(yg, xg, cu(go))
    assert_close(yg, yr, 2e-6, "pixelnorm out")
    assert_close(gg, gr, 5e-6, "pixelnorm grad")
    # degenerate row: all zeros -> 0 * rsqrt(1e-8) = 0, gradient = g * 1e4 (finite)
    z = torch.zeros(2, shape[1], device=DEV, requires_grad=True)
    yz = M.PixelNorm()(z)
    (gz,) = torch.autograd.grad(yz, z, torch.ones_like(yz))
    assert float(yz.abs().max()) == 0.0 and torch.isfinite(gz).all()
    assert_close(gz, torch.full_like(gz, 1e4).cpu(), 1e-5, "pixelnorm grad at 0")


@pytest.mark.parametrize("int_step", [False, True], ids=["tensor_step", "int_step_torch16"])
def test_graphed_kd_step_resumes_from_saved_optimizer_state_and_invalidates_frozen_caches(int_step):
    """(int_step: the checkpoint stores Adam's `step` as a Python int, as torch 1.6 — the reference's pin — writes it; advisor
    round 3: it must reach the captured graph's device-side counter, or the bias correction restarts at t = 1.)

    Advisor round 2: (1) a GraphedKDStep resumed from a checkpoint's Adam state (checkpoint.restore_optimizers ->
    load_optim_state copies INTO the tensors the captured graph holds) continues exactly like the uninterrupted run;
    (2) a frozen (no-grad) use of the student between replays sees the replayed weights, not stale packed ones."""
    from cagc import checkpoint as ck
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")

    def build():
        student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
        student.load_state_dict(sub(g, "student_sd/"), strict=True)
        teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
        teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
        disc = M.Discriminator(32)
        disc.load_state_dict(ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
        return student.to(DEV), teacher.to(DEV), disc.to(DEV)

    B = g["mask"].shape[0]
    steps = meta["steps"]

    def inputs(st, student):
        p = f"step{st['step']}/"
        n = student.num_layers
        return ([cu(g[p + f"z{i}"]) for i in range(st["n_z"])], st["inject_index"], cu(g["mask"]),
                [cu(g[p + f"student_noise{i}"]) for i in range(n)], [cu(g[p + f"teacher_noise{i}"]) for i in range(n)])

    # uninterrupted: two replays
    s0, t0, d0 = build()
    run0 = kd.GraphedKDStep(s0, t0, d0, B, cu(g["mask"]), random_noise=False, latent=24)
    run0.g_step(*inputs(steps[0], s0))
    saved = {"g": {k: v.detach().clone() for k, v in s0.state_dict().items()},
             "g_optim": {"state": {i: {k: ((int(v.item()) if (int_step and k == "step") else v.detach().clone()) if torch.is_tensor(v) else v)
                                       for k, v in st.items()}
                                   for i, st in run0.optim.state_dict()["state"].items()},
                         "param_groups": run0.optim.state_dict()["param_groups"]}}
    # frozen use between replays: eval forward must use the UPDATED weights (caches keyed on _version would be stale)
    z = cu(g["step0/z0"])
    with torch.no_grad():
        kd.requires_grad(s0, False)
        img_frozen = s0([z], randomize_noise=False)
        kd.requires_grad(s0, True)
        img_live = s0([z], randomize_noise=False)
    assert_close(img_frozen, img_live, 1e-6, "frozen forward after a graph replay")
    run0.g_step(*inputs(steps[1], s0))
    # resumed: fresh objects, weights + Adam state restored, one replay
    s1, t1, d1 = build()
    s1.load_state_dict(saved["g"])
    run1 = kd.GraphedKDStep(s1, t1, d1, B, cu(g["mask"]), random_noise=False, latent=24)
    ck.restore_optimizers(saved, run1)
    run1.g_step(*inputs(steps[1], s1))
    torch.cuda.synchronize()
    p0, p1 = dict(s0.named_parameters()), dict(s1.named_parameters())
    for k in p0:
        assert_close(p1[k].detach(), p0[k].detach(), 2e-4, "resumed graph replay param " + k)   # atomics order differs run to run
    st0 = run0.optim.state_dict()["state"]
    st1 = run1.optim.state_dict()["state"]
    for i in st0:
        assert_close(st1[i]["exp_avg_sq"], st0[i]["exp_avg_sq"], 2e-3, f"resumed Adam second moment {i}")
        assert float(st1[i]["step"]) == float(st0[i]["step"]) == 2.0
    # advisor round 4: a checkpoint WRITTEN from the graphed step must not carry its aliased state (one shared `step` tensor, moments
    # that are slices of one buffer) — a plain Adam resumed from it would advance the shared counter once per parameter per step
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = ck.save_checkpoint(td, 2, s0, d0, s0, g_optim=run0)
        raw = torch.load(path)
    steps_saved = [st["step"] for st in raw["g_optim"]["state"].values()]
    assert len({t.data_ptr() for t in steps_saved}) == len(steps_saved), "every parameter owns its step counter in the checkpoint"
    s2, t2, d2 = build()
    s2.load_state_dict(raw["g"])
    eager = kd.KDStep(s2, t2, d2, latent=24)
    ck.restore_optimizers(raw, eager.optim)
    eager.g_step(*inputs(steps[1], s2))
    assert all(float(st["step"]) == 3.0 for st in eager.optim.state_dict()["state"].values()), "plain Adam resumed from the graphed step's checkpoint: t + 1"
    # hyper-parameters live inside the captured Adam graph: a checkpoint written with others is refused, not silently ignored
    bad = {"g_optim": {"state": saved["g_optim"]["state"], "param_groups": [dict(pg, lr=pg["lr"] * 2) for pg in saved["g_optim"]["param_groups"]]}}
    with pytest.raises(ValueError, match="lr"):
        run1.load_optim_state(bad["g_optim"])
