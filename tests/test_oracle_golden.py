"""Pin the oracle (oracle/*.py, torch-fp32 CPU restatement) to golden vectors captured from the
reference's own CPU path by oracle/gen_golden.py.  Tolerance 1e-5 rel (same torch CPU kernels, only the
op order differs); the parity gate for the HIP path is 1e-3 (BASELINE.json north_star)."""
import torch

from oracle import ref_kd, ref_model, ref_ops
from _util import assert_close, assert_grad_matches_sample, load_json, load_npz, sub

TOL = 1e-5


def test_fused_leaky_relu_first_and_second_order():
    g = load_npz("fused_act")
    for tag in ("2d", "4d"):
        for bias in ("b", "nb"):
            k = f"{tag}_{bias}"
            x = g[k + "_x"].clone().requires_grad_(True)
            b = g[k + "_b"].clone().requires_grad_(True) if bias == "b" else None
            go = g[k + "_go"].clone().requires_grad_(True)
            y = ref_ops.fused_leaky_relu_ref(x, b)
            assert_close(y, g[k + "_y"], TOL, k + " y")
            ins = [x] + ([b] if b is not None else [])
            grads = torch.autograd.grad(y, ins, go, create_graph=True)
            assert_close(grads[0], g[k + "_gx"], TOL, k + " gx")
            if b is not None:
                assert_close(grads[1], g[k + "_gb"], TOL, k + " gb")
            (ggo,) = torch.autograd.grad(grads[0], go, g[k + "_ggi"])
            assert_close(ggo, g[k + "_ggo"], TOL, k + " ggo")


def test_upfirdn2d_all_reachable_configs():
    g = load_npz("upfirdn2d")
    for c in load_json("upfirdn2d_cases"):
        n = c["name"]
        x = g[n + "_x"].clone().requires_grad_(True)
        y = ref_ops.upfirdn2d_ref(x, g[n + "_k"], up=c["up"], down=c["down"], pad=tuple(c["pad"]))
        assert_close(y, g[n + "_y"], TOL, n + " y")
        (gx,) = torch.autograd.grad(y, x, g[n + "_go"])
        assert_close(gx, g[n + "_gx"], TOL, n + " gx")


def test_modulated_conv_plain_up_down_rgb():
    g = load_npz("modconv")
    for c in load_json("modconv_cases"):
        n = c["name"]
        leaves = {k: g[f"{n}_{k}"].clone().requires_grad_(True) for k in ("x", "w", "weight", "mod_weight", "mod_bias")}
        y, s = ref_ops.modulated_conv2d_ref(leaves["x"], leaves["w"], leaves["weight"], leaves["mod_weight"],
                                            leaves["mod_bias"], demodulate=c.get("demodulate", True),
                                            upsample=c.get("upsample", False), downsample=c.get("downsample", False))
        assert_close(y, g[n + "_y"], TOL, n + " y")
        assert_close(s, g[n + "_s"], TOL, n + " s")
        grads = torch.autograd.grad(y, list(leaves.values()), g[n + "_go"])
        for (k, _), gr in zip(leaves.items(), grads):
            assert_close(gr, g[f"{n}_g{k}"], 2e-5, f"{n} g{k}")


def _tiny():
    g = load_npz("generator_tiny")
    sd = sub(g, "sd/")
    return g, sd


def test_generator_rgb_list_and_style_scalars():
    g, sd = _tiny()
    rgbs, scal = ref_model.generator_forward_ref(sd, [g["z0"]], randomize_noise=False, return_rgb_list=True,
                                                 return_style_scalars=True)
    assert len(scal) == g["a_n_styles"]
    for i, r in enumerate(rgbs):
        assert_close(r, g[f"a_rgb{i}"], TOL, f"rgb{i}")
    for i, s in enumerate(scal):
        assert_close(s, g[f"a_style{i}"], TOL, f"style{i}")


def test_generator_all_param_grads():
    g, sd = _tiny()
    keys = load_json("generator_tiny_keys")["keys"]
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    img = ref_model.generator_forward_ref(sd, [g["z0"]], randomize_noise=False)
    assert_close(img, g["b_img"], TOL, "img")
    loss = img.abs().mean()
    assert abs(loss.item() - float(g["b_loss"])) < 1e-6
    gold = sub(g, "b_grad/")
    names = [k for k in gold]
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    for k, gr in zip(names, grads):
        gr = torch.zeros_like(sd[k]) if gr is None else gr
        assert_close(gr, gold[k], 5e-5, "grad " + k)
    assert [k for k, _ in keys] == list(sub(g, "sd/").keys())


def test_generator_mixing_truncation_noise():
    g, sd = _tiny()
    assert_close(ref_model.generator_forward_ref(sd, [g["z0"], g["z1"]], inject_index=3, randomize_noise=False),
                 g["c_img"], TOL, "mixing")
    w0 = ref_model.mapping_ref(sd, g["z0"])
    assert_close(w0, g["d_w0"], TOL, "mapping")
    assert_close(ref_model.generator_forward_ref(sd, latents=[g["d_w0"]], truncation=0.7,
                                                 truncation_latent=g["d_mean_w"], randomize_noise=False),
                 g["d_img"], TOL, "truncation")
    n_layers = load_json("generator_tiny_keys")["num_layers"]
    noise = [g[f"e_noise{i}"] for i in range(n_layers)]
    assert_close(ref_model.generator_forward_ref(sd, [g["z0"]], noise=noise), g["e_img"], TOL, "explicit noise")


def test_generator512_mapping_modulation_at_the_real_latent_width():
    """style_dim = 512 (reference model.py:137-171,421-430): weights from the seeded recipe the fixture was generated with."""
    g = load_npz("generator512")
    meta = load_json("generator512_keys")
    cfg = meta["config"]
    sd = ref_model.regenerate_generator_state_dict(meta["keys"], cfg["seed"])
    chk = {k: i for i, (k, _) in enumerate(meta["keys"])}
    for k, v in sd.items():      # the recipe reproduces what gen_golden.py loaded into the reference
        assert abs(float(v.double().sum()) - float(g["sd_checksum"][chk[k]])) <= 1e-6 * max(1.0, float(g["sd_abs_checksum"][chk[k]])), k
    assert_close(ref_model.mapping_ref(sd, g["z0"]), g["a_w0"], TOL, "mapping")
    rgbs, scal = ref_model.generator_forward_ref(sd, [g["z0"]], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    assert len(scal) == g["a_n_styles"]
    for i, r in enumerate(rgbs):
        assert_close(r, g[f"a_rgb{i}"], TOL, f"rgb{i}")
    for i, s_ in enumerate(scal):
        assert_close(s_, g[f"a_style{i}"], TOL, f"style{i}")
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    img = ref_model.generator_forward_ref(leaves, [g["z0"], g["z1"]], inject_index=3, randomize_noise=False)
    assert_close(img, g["b_img"], TOL, "mixing image")
    names = list(sub(g, "b_grad/"))
    grads = torch.autograd.grad(img.abs().mean(), [leaves[k] for k in names], allow_unused=True)
    for k, gr in zip(names, grads):
        gr = torch.zeros_like(leaves[k]) if gr is None else gr
        assert_grad_matches_sample(gr, g["b_grad/" + k], g["b_gsum/" + k], cfg, 5e-5, "grad " + k)


def test_generator_path_length_double_backward():
    g, sd = _tiny()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    img, latent = ref_model.generator_forward_ref(sd, [g["z0"]], randomize_noise=False, return_latent=True)
    pl = ref_model.path_lengths_ref(img, latent, g["f_pl_noise"])
    assert_close(pl, g["f_path_lengths"], TOL, "path lengths")
    loss = (pl - 0.37).pow(2).mean()
    gold = sub(g, "f_grad/")
    names = list(gold)
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    for k, gr in zip(names, grads):
        gr = torch.zeros_like(sd[k]) if gr is None else gr
        assert_close(gr, gold[k], 1e-4, "pl grad " + k)


def test_discriminator_forward_and_input_grad():
    g = load_npz("discriminator32")
    sd = ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["seed"])
    chk = torch.tensor([float(v.double().sum()) for v in sd.values()], dtype=torch.float64)
    assert torch.allclose(chk, g["checksum"].double(), rtol=0, atol=1e-6), "regenerated D differs from the reference's"
    x = g["x"].clone().requires_grad_(True)
    y = ref_model.discriminator_forward_ref(sd, x)
    assert_close(y, g["y"], 2e-5, "D out")
    (gx,) = torch.autograd.grad(torch.nn.functional.softplus(-y).mean(), x)
    assert_close(gx, g["gx"], 5e-5, "D input grad")


def test_kd_generator_step_losses_grads_and_adam():
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    d_sd = ref_model.regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"])
    chk = torch.tensor([float(v.double().sum()) for v in d_sd.values()], dtype=torch.float64)
    assert torch.allclose(chk, g["d_checksum"].double(), rtol=0, atol=1e-6)
    student = sub(g, "student_sd/")
    teacher = sub(g, "teacher_sd/")
    adam = {}
    for st in meta["steps"]:
        p = f"step{st['step']}/"
        gold_grads = sub(g, p + "grad/")
        names = list(gold_grads)
        leaves = {k: student[k].clone().requires_grad_(True) for k in names}
        sd = dict(student)
        sd.update(leaves)
        zs = [g[p + f"z{i}"] for i in range(st["n_z"])]
        nl = len([k for k in g if k.startswith(p + "student_noise")])
        g_loss, kd_l1, _ = ref_kd.kd_generator_losses_ref(
            sd, teacher, d_sd, zs, st["inject_index"], g["mask"],
            student_noise=[g[p + f"student_noise{i}"] for i in range(nl)],
            teacher_noise=[g[p + f"teacher_noise{i}"] for i in range(nl)])
        assert abs(g_loss.item() - float(g[p + "g_loss"])) < 2e-5 * max(1, abs(float(g[p + "g_loss"])))
        assert abs(kd_l1.item() - float(g[p + "kd_l1_loss"])) < 2e-5 * max(1, abs(float(g[p + "kd_l1_loss"])))
        assert float(g[p + "kd_lpips_loss"]) == 0.0
        grads = torch.autograd.grad(g_loss + kd_l1, [leaves[k] for k in names], allow_unused=True)
        gd = {}
        for k, gr in zip(names, grads):
            gd[k] = torch.zeros_like(leaves[k]) if gr is None else gr
            assert_close(gd[k], gold_grads[k], 2e-4, f"step{st['step']} grad {k}")
        new = ref_kd.adam_step_ref({k: student[k] for k in names}, gd, adam, meta["lr"], meta["betas"])
        for k in names:
            assert_close(new[k], g[p + "param_after/" + k], 2e-5, f"step{st['step']} adam {k}")
            student[k] = g[p + "param_after/" + k]     # follow the reference trajectory exactly
