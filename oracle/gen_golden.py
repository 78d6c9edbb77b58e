#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own CPU path.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference, which
does not exist on the GPU box); what travels is the *data* it writes (tests/golden/*.npz,
*.json) — inputs and expected outputs — never reference source.

How the reference is imported: bytecode writing off, the CUDA JIT loader
(`torch.utils.cpp_extension.load`, used at import by op/fused_act.py:11-17 and
op/upfirdn2d.py:10-16) replaced by a no-op so nothing is hipified or written into
/root/reference, torchvision stubbed (only `make_grid` is ever used from it).  Every tensor
stays on CPU, so the reference takes its own CPU branches (op/fused_act.py:105-116,
op/upfirdn2d.py:146-149,159-200).

The KD-step golden executes the reference's *own* `G_Loss_BackProp`, `KD_loss`,
`g_nonsaturating_loss`, `index_aware_mixing_noise`, `make_noise`, `requires_grad`
(train.py:119-121,145-184,203-237,280-308).  train.py cannot be imported (argparse and
`cuda:0` at module scope), so those FunctionDefs are lifted out of its AST and executed in a
namespace built here; nothing of their text is stored.

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import ast
import json
import os
import random
import sys
import types
from unittest import mock

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("CAGC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.utils.cpp_extension as _ce  # noqa: E402

_ce.load = lambda *a, **k: None  # CPU branches never touch the extension handle
for _m in ("torchvision", "torchvision.utils", "torchvision.transforms"):
    sys.modules[_m] = mock.MagicMock()

import torch.nn.functional as F  # noqa: E402
from torch import autograd  # noqa: E402

import model as ref_model  # noqa: E402  (reference model.py)
from op import fused_leaky_relu as ref_flrelu, upfirdn2d as ref_upfirdn2d  # noqa: E402
from Util.network_util import Get_Network_Shape, Build_Generator_From_Dict  # noqa: E402
from Util.mask_util import Mask_the_Generator  # noqa: E402
from Util.pruning_util import Get_Uniform_RmveList, Generate_Prune_Mask_List  # noqa: E402
from Util.content_aware_pruning import Batch_Img_Parsing, Get_Masked_Tensor  # noqa: E402
from Util import Calculators  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path), "B")


# ----------------------------------------------------------------------------------------------
# 1. fused_leaky_relu  (op/fused_act.py:104-119)
# ----------------------------------------------------------------------------------------------
def gold_fused_act():
    g = torch.Generator().manual_seed(11)
    out = {}
    for tag, shape in (("2d", (5, 7)), ("4d", (3, 7, 6, 5))):
        for with_bias in (True, False):
            x = torch.randn(*shape, generator=g, requires_grad=True)
            b = torch.randn(shape[1], generator=g, requires_grad=True) if with_bias else None
            go = torch.randn(*shape, generator=g, requires_grad=True)
            ggi = torch.randn(*shape, generator=g)
            y = ref_flrelu(x, b)
            ins = [x] + ([b] if with_bias else [])
            grads = autograd.grad(y, ins, go, create_graph=True)
            # second order: d(grad_input)/d(grad_output) contracted with ggi
            (ggo,) = autograd.grad(grads[0], go, ggi, retain_graph=True)
            k = f"{tag}_{'b' if with_bias else 'nb'}"
            out[k + "_x"] = x
            out[k + "_go"] = go
            out[k + "_ggi"] = ggi
            out[k + "_y"] = y
            out[k + "_gx"] = grads[0]
            out[k + "_ggo"] = ggo
            if with_bias:
                out[k + "_b"] = b
                out[k + "_gb"] = grads[1]
    save("fused_act", **out)


# ----------------------------------------------------------------------------------------------
# 2. upfirdn2d  (op/upfirdn2d.py:145-200), every configuration of SURVEY.md App. B + edge cases
# ----------------------------------------------------------------------------------------------
UPFIRDN_CASES = [
    # name, (N,C,H,W), ktaps, gain, up, down, pad
    ("blur_up_pad11", (2, 3, 9, 9), [1, 3, 3, 1], 4.0, 1, 1, (1, 1)),     # ModulatedConv2d up blur
    ("skip_up2_pad21", (2, 3, 9, 9), [1, 3, 3, 1], 4.0, 2, 1, (2, 1)),    # ToRGB.upsample
    ("down2_pad11", (2, 3, 18, 18), [1, 3, 3, 1], 1.0, 1, 2, (1, 1)),     # Downsample / bwd of up2
    ("blur_pad22", (2, 3, 8, 8), [1, 3, 3, 1], 1.0, 1, 1, (2, 2)),        # D ConvLayer 3x3 down
    ("blur_pad11_g1", (2, 3, 8, 8), [1, 3, 3, 1], 1.0, 1, 1, (1, 1)),     # D ConvLayer 1x1 skip
    ("negpad", (1, 2, 10, 12), [1, 3, 3, 1], 1.0, 1, 1, (-1, 2)),         # negative pad crops
    ("odd_rect_up2", (1, 2, 5, 7), [1, 3, 3, 1], 4.0, 2, 1, (2, 1)),      # non-square, odd
    ("k2_up2", (1, 2, 6, 6), [1, 1], 4.0, 2, 1, (1, 0)),                  # 2x2 kernel (CUDA mode 4)
    ("k6_generic", (1, 2, 9, 11), [1, 2, 3, 3, 2, 1], 1.0, 1, 1, (3, 2)),  # >4 taps: "large" path
    ("up2_down2", (1, 2, 7, 7), [1, 3, 3, 1], 1.0, 2, 2, (2, 1)),         # mixed up/down: "large" path
    ("asym_kernel", (1, 2, 8, 8), None, 1.0, 1, 1, (2, 1)),               # non-symmetric 3x4 kernel: flip check
]


def gold_upfirdn2d():
    g = torch.Generator().manual_seed(12)
    out = {}
    meta = []
    for name, shape, taps, gain, up, down, pad in UPFIRDN_CASES:
        x = torch.randn(*shape, generator=g, requires_grad=True)
        if taps is None:
            k = torch.randn(3, 4, generator=g)
        else:
            k = ref_model.make_kernel(taps) * gain
        y = ref_upfirdn2d(x, k, up=up, down=down, pad=pad)
        go = torch.randn(*y.shape, generator=g)
        (gx,) = autograd.grad(y, x, go)
        out[name + "_x"], out[name + "_k"], out[name + "_y"] = x, k, y
        out[name + "_go"], out[name + "_gx"] = go, gx
        meta.append(dict(name=name, up=up, down=down, pad=list(pad)))
    save("upfirdn2d", **out)
    with open(os.path.join(OUT, "upfirdn2d_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ----------------------------------------------------------------------------------------------
# 3. ModulatedConv2d  (model.py:186-289)
# ----------------------------------------------------------------------------------------------
MODCONV_CASES = [
    # name, cin, cout, k, H, W, kwargs
    ("plain_7_5", 7, 5, 3, 6, 6, {}),
    ("plain_39_39", 39, 39, 3, 8, 8, {}),
    ("plain_77_39_rect", 77, 39, 3, 5, 9, {}),
    ("up_7_5", 7, 5, 3, 4, 4, dict(upsample=True)),
    ("up_77_39", 77, 39, 3, 8, 8, dict(upsample=True)),
    ("down_7_5", 7, 5, 3, 8, 8, dict(downsample=True)),
    ("rgb_39_3", 39, 3, 1, 8, 8, dict(demodulate=False)),
    ("rgb_7_3", 7, 3, 1, 5, 5, dict(demodulate=False)),
]
STYLE_DIM_SMALL = 24


def gold_modconv():
    out = {}
    meta = []
    for idx, (name, cin, cout, ks, H, W, kw) in enumerate(MODCONV_CASES):
        torch.manual_seed(100 + idx)
        m = ref_model.ModulatedConv2d(cin, cout, ks, STYLE_DIM_SMALL, **kw)
        with torch.no_grad():
            m.modulation.bias.add_(0.3 * torch.randn_like(m.modulation.bias))
        B = 3
        x = torch.randn(B, cin, H, W, requires_grad=True)
        w = torch.randn(B, STYLE_DIM_SMALL, requires_grad=True)
        y, s = m(x, w, return_style_scalars=True)
        go = torch.randn_like(y)
        params = [m.weight, m.modulation.weight, m.modulation.bias]
        grads = autograd.grad(y, [x, w] + params, go)
        out[name + "_weight"] = m.weight
        out[name + "_mod_weight"] = m.modulation.weight
        out[name + "_mod_bias"] = m.modulation.bias
        out[name + "_x"], out[name + "_w"], out[name + "_y"], out[name + "_s"] = x, w, y, s
        out[name + "_go"] = go
        out[name + "_gx"], out[name + "_gw"] = grads[0], grads[1]
        out[name + "_gweight"], out[name + "_gmod_weight"], out[name + "_gmod_bias"] = grads[2:]
        meta.append(dict(name=name, cin=cin, cout=cout, k=ks, H=H, W=W, style_dim=STYLE_DIM_SMALL, **kw))
    save("modconv", **out)
    with open(os.path.join(OUT, "modconv_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ----------------------------------------------------------------------------------------------
# 4. tiny Generator  (model.py:398-666)
# ----------------------------------------------------------------------------------------------
TINY = dict(size=32, style_dim=24, n_mlp=2, shape=[11, 11, 7, 7, 5, 5, 3, 3])


def make_tiny_generator(seed, shape=None):
    torch.manual_seed(seed)
    gnet = ref_model.Generator(TINY["size"], TINY["style_dim"], TINY["n_mlp"],
                               generator_net_shape=shape or TINY["shape"])
    with torch.no_grad():  # noise weights init to 0 (model.py:296) would hide the noise path
        for n, p in gnet.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1 + 0.05 * torch.randn(()).item())
            if n.endswith("activate.bias") or (n.startswith("to_rgb") and n.endswith(".bias") and p.ndim == 4):
                p.copy_(0.1 * torch.randn_like(p))
    return gnet


def sd_arrays(prefix, sd):
    return {prefix + k: v for k, v in sd.items()}


def gold_generator():
    gnet = make_tiny_generator(200)
    sd = {k: v.clone() for k, v in gnet.state_dict().items()}
    out = sd_arrays("sd/", sd)
    with open(os.path.join(OUT, "generator_tiny_keys.json"), "w") as f:
        json.dump(dict(config=TINY, keys=[[k, list(v.shape)] for k, v in sd.items()],
                       n_latent=gnet.n_latent, num_layers=gnet.num_layers), f, indent=1)
    gtor = torch.Generator().manual_seed(201)
    B = 3
    z0 = torch.randn(B, TINY["style_dim"], generator=gtor)
    z1 = torch.randn(B, TINY["style_dim"], generator=gtor)
    out["z0"], out["z1"] = z0, z1

    # (a) single latent, fixed noise buffers, rgb list + style scalars
    rgbs, styles = gnet([z0], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    for i, r in enumerate(rgbs):
        out[f"a_rgb{i}"] = r
    for i, s in enumerate(styles):
        out[f"a_style{i}"] = s
    out["a_n_styles"] = np.int64(len(styles))
    # (b) all parameter grads for L = |img|.mean()
    gnet.zero_grad()
    img = gnet([z0], randomize_noise=False)
    loss = img.abs().mean()
    loss.backward()
    out["b_img"], out["b_loss"] = img, loss
    for n, p in gnet.named_parameters():
        out["b_grad/" + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    # (c) style mixing at inject_index = 3
    out["c_img"] = gnet([z0, z1], inject_index=3, randomize_noise=False)
    # (d) truncation + input_is_latent
    with torch.no_grad():
        w0 = gnet.get_latent(z0)
        mean_w = w0.mean(0, keepdim=True)
    out["d_w0"], out["d_mean_w"] = w0, mean_w
    out["d_img"] = gnet(None, latent_styles=[w0], input_is_latent=True, truncation=0.7,
                        truncation_latent=mean_w, randomize_noise=False)
    # (e) explicit per-sample noise list (what randomize_noise=True draws, but recorded)
    noises = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gtor)
              for i in range(gnet.num_layers)]
    for i, n in enumerate(noises):
        out[f"e_noise{i}"] = n
    out["e_img"] = gnet([z0], noise=noises)
    # (f) path-length double backward (model.py:661-666, train.py:310-338)
    gnet.zero_grad()
    pl_noise = torch.randn(B, 3, TINY["size"], TINY["size"], generator=gtor)
    out["f_pl_noise"] = pl_noise
    with mock.patch.object(torch, "randn_like", lambda t: pl_noise):
        img_f, path_lengths = gnet([z0], PPL_regularize=True, randomize_noise=False)
    out["f_img"], out["f_path_lengths"] = img_f, path_lengths
    pl_loss = (path_lengths - 0.37).pow(2).mean()
    pl_loss.backward()
    out["f_loss"] = pl_loss
    for n, p in gnet.named_parameters():
        out["f_grad/" + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    save("generator_tiny", **out)


# ----------------------------------------------------------------------------------------------
# 4b. tiny Generator with the REAL latent width (style_dim = 512, model.py:137-171,421-430): the fixture that reaches the product's
#     one-launch mapping / modulation kernels (csrc/mapping.hip, modbank.hip serve in_dim % 512 == 0 only).  The weights come from
#     the seeded recipe oracle/ref_model.regenerate_generator_state_dict (the mapping network alone is 2 MB); large gradients are
#     stored as (sum, abs-sum) + a strided sample.
# ----------------------------------------------------------------------------------------------
G512 = dict(size=32, style_dim=512, n_mlp=2, shape=[8, 8, 6, 6, 4, 4, 3, 3], seed=210, sample_stride=97, full_below=4097)


def sampled(t, cfg=G512):
    """what the fixture keeps of a tensor: everything when small, else a strided sample (+ checksums stored beside it)"""
    flat = t.detach().reshape(-1)
    return flat if flat.numel() < cfg["full_below"] else flat[::cfg["sample_stride"]].clone()


def gold_generator512():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.ref_model import regenerate_generator_state_dict
    torch.manual_seed(G512["seed"])
    gnet = ref_model.Generator(G512["size"], G512["style_dim"], G512["n_mlp"], generator_net_shape=G512["shape"])
    keys = [[k, list(v.shape)] for k, v in gnet.state_dict().items()]
    missing, unexpected = gnet.load_state_dict(regenerate_generator_state_dict(keys, G512["seed"]), strict=False)
    assert not unexpected and all(k.endswith("kernel") for k in missing), (missing, unexpected)
    sd = gnet.state_dict()
    with open(os.path.join(OUT, "generator512_keys.json"), "w") as f:
        json.dump(dict(config=G512, keys=keys, n_latent=gnet.n_latent, num_layers=gnet.num_layers), f, indent=1)
    out = {"sd_checksum": np.array([float(v.double().sum()) for v in sd.values()]),
           "sd_abs_checksum": np.array([float(v.double().abs().sum()) for v in sd.values()])}
    gtor = torch.Generator().manual_seed(G512["seed"] + 1)
    B = 2
    z0 = torch.randn(B, G512["style_dim"], generator=gtor)
    z1 = torch.randn(B, G512["style_dim"], generator=gtor)
    out["z0"], out["z1"] = z0, z1
    # (a) mapping network alone, rgb list + style scalars
    out["a_w0"] = gnet.get_latent(z0)
    rgbs, styles = gnet([z0], randomize_noise=False, return_rgb_list=True, return_style_scalars=True)
    for i, r in enumerate(rgbs):
        out[f"a_rgb{i}"] = r
    for i, s_ in enumerate(styles):
        out[f"a_style{i}"] = s_
    out["a_n_styles"] = np.int64(len(styles))
    # (b) style mixing at inject_index = 3, every parameter gradient of L = |img|.mean()
    gnet.zero_grad()
    img = gnet([z0, z1], inject_index=3, randomize_noise=False)
    loss = img.abs().mean()
    loss.backward()
    out["b_img"], out["b_loss"] = img, loss
    for n, p in gnet.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        out["b_grad/" + n] = sampled(gr)
        out["b_gsum/" + n] = np.array([float(gr.double().sum()), float(gr.double().abs().sum()), float(gr.abs().max())])
    save("generator512", **out)


# ----------------------------------------------------------------------------------------------
# 5. KD generator step  (train.py:280-308 via AST-lifted functions)
# ----------------------------------------------------------------------------------------------
def lift_train_functions(names, namespace):
    with open(os.path.join(REF, "train.py")) as f:
        tree = ast.parse(f.read())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in picked} == set(names), "train.py layout changed"
    code = compile(ast.Module(body=picked, type_ignores=[]), "<lifted from reference train.py>", "exec")
    exec(code, namespace)
    return namespace


class RecordingNoise:
    """Replace NoiseInjection.forward (model.py:298-303) for the duration of the KD golden so the
    per-layer N(0,1) draws of `randomize_noise=True` are recorded (same semantics: fresh normal
    noise [B,1,H,W] whenever the caller passed noise=None)."""

    def __init__(self):
        self.log = []

    def __call__(self, module, image, noise=None):
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        self.log.append(noise.detach().clone())
        return image + module.weight * noise


def gold_kd_step():
    student = make_tiny_generator(300, shape=[5, 5, 4, 4, 3, 3, 2, 2])
    teacher = make_tiny_generator(301)  # full tiny shape
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    torch.manual_seed(302)
    # Discriminator(32): channels dict gives 512 everywhere at <=32 px — too big for a fixture.
    # Build with channel_multiplier irrelevant; shrink by monkeypatching the class-level table is
    # not possible (local dict), so build size-32 D and slice nothing: instead use size 32 with
    # the stock 512 channels but keep only outputs/grads in the fixture and regenerate D weights
    # from a seed on the test side.  D weights are therefore stored as a seed + checksum only.
    disc = ref_model.Discriminator(TINY["size"])
    d_sd = disc.state_dict()

    ns = dict(torch=torch, F=F, autograd=autograd, random=random, device="cpu",
              Batch_Img_Parsing=Batch_Img_Parsing, Get_Masked_Tensor=Get_Masked_Tensor,
              train_hyperparams=types.SimpleNamespace(LPIPS_IMAGE_SIZE=256))
    lift_train_functions(["requires_grad", "KD_loss", "g_nonsaturating_loss", "make_noise",
                          "index_aware_mixing_noise", "G_Loss_BackProp"], ns)

    B = 4
    # synthetic BiSeNet stand-in: fixed 19-class logits at 512x512 whose argmax is a known map
    # (classes 0 = background and 16 = cloth are masked OUT by Get_Masked_Tensor, :102).
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        r = ((yy - 256 - 10 * i) / 200.0) ** 2 + ((xx - 256 + 7 * i) / 150.0) ** 2
        cls[i][r < 1.0] = 1 + i          # face-ish classes: kept
        cls[i][(yy > 440)] = 16          # cloth: masked out
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).float()

    def fake_parsing_net(x):
        assert x.shape == (B, 3, 512, 512)
        return (logits,)

    args = types.SimpleNamespace(batch_size=B, latent=TINY["style_dim"], mixing=0.9,
                                 n_latent=student.n_latent, kd_mode="Output_Only", kd_l1_lambda=3,
                                 kd_lpips_lambda=3, size=TINY["size"])
    lr, c = 0.002, 4 / 5   # train.py:528-532
    g_optim = torch.optim.Adam(student.parameters(), lr=lr * c, betas=(0.0 ** c, 0.99 ** c))

    rec = RecordingNoise()
    captured = {}
    orig_fwd = ref_model.Generator.forward

    def capturing_forward(self, noise_z, *a, **k):
        captured.setdefault("calls", []).append(([t.detach().clone() for t in noise_z], k.get("inject_index")))
        return orig_fwd(self, noise_z, *a, **k)

    out = {}
    sd0 = {k: v.clone() for k, v in student.state_dict().items()}
    out.update(sd_arrays("student_sd/", sd0))
    out.update(sd_arrays("teacher_sd/", teacher.state_dict()))
    steps = []
    with mock.patch.object(ref_model.NoiseInjection, "forward", lambda self, image, noise=None: rec(self, image, noise)), \
            mock.patch.object(ref_model.Generator, "forward", capturing_forward):
        for step in range(2):
            random.seed((400, 417)[step])  # 400 -> style mixing with an inject_index; 417 -> the no-mixing branch
            torch.manual_seed(500 + step)
            rec.log.clear()
            captured.clear()
            loss_dict = {}
            ns["G_Loss_BackProp"](student, disc, args, "cpu", loss_dict, g_optim, teacher, None, fake_parsing_net)
            (zs, inj), (zt, injt) = captured["calls"]
            assert inj == injt and all(torch.equal(a, b) for a, b in zip(zs, zt))
            nl = student.num_layers
            assert len(rec.log) == 2 * nl
            p = f"step{step}/"
            out[p + "n_z"] = np.int64(len(zs))
            for i, z in enumerate(zs):
                out[p + f"z{i}"] = z
            out[p + "inject_index"] = np.int64(-1 if inj is None else inj)
            for i in range(nl):
                out[p + f"student_noise{i}"] = rec.log[i]
                out[p + f"teacher_noise{i}"] = rec.log[nl + i]
            out[p + "g_loss"] = loss_dict["g"]
            out[p + "kd_l1_loss"] = loss_dict["kd_l1_loss"]
            out[p + "kd_lpips_loss"] = loss_dict["kd_lpips_loss"]
            for n, prm in student.named_parameters():
                out[p + "grad/" + n] = prm.grad.detach().clone()
                out[p + "param_after/" + n] = prm.detach().clone()
            steps.append(dict(step=step, n_z=len(zs), inject_index=inj))
    # the {0,1} mask the reference derived from the parsing at image resolution (:102-107)
    mask = ((cls > 0) * (cls != 16)).unsqueeze(0).float()
    mask = (F.interpolate(mask, scale_factor=TINY["size"] / 512, mode="bilinear", align_corners=False).squeeze() > 0.5).float()
    out["mask"] = mask.view(B, 1, TINY["size"], TINY["size"])
    out["d_seed"] = np.int64(302)
    # D weights: too large to commit (29 MB); store a few checksums so the test can prove its
    # regenerated D (same seed, same constructor order) is the same tensor set.
    out["d_checksum"] = np.array([float(v.double().sum()) for v in d_sd.values()])
    out["d_abs_checksum"] = np.array([float(v.double().abs().sum()) for v in d_sd.values()])
    with open(os.path.join(OUT, "discriminator32_keys.json"), "w") as f:
        json.dump([[k, list(v.shape)] for k, v in d_sd.items()], f, indent=1)
    save("kd_step_tiny", **out)
    with open(os.path.join(OUT, "kd_step_tiny_meta.json"), "w") as f:
        json.dump(dict(steps=steps, batch=B, student_shape=[5, 5, 4, 4, 3, 3, 2, 2], teacher_shape=TINY["shape"],
                       lr=lr * c, betas=[0.0 ** c, 0.99 ** c], kd_l1_lambda=3), f, indent=1)


# ----------------------------------------------------------------------------------------------
# 6. Discriminator forward + input grad (model.py:740-798), small fixture via seed-regenerated weights
# ----------------------------------------------------------------------------------------------
def gold_discriminator():
    torch.manual_seed(600)
    disc = ref_model.Discriminator(TINY["size"])
    g = torch.Generator().manual_seed(601)
    x = torch.randn(4, 3, TINY["size"], TINY["size"], generator=g, requires_grad=True)
    y = disc(x)
    (gx,) = autograd.grad(F.softplus(-y).mean(), x)
    save("discriminator32", x=x, y=y, gx=gx, seed=np.int64(600),
         checksum=np.array([float(v.double().sum()) for v in disc.state_dict().values()]))


# ----------------------------------------------------------------------------------------------
# 7. state-dict contract + prune chain + MAC KATs (Util/network_util.py, mask_util.py, Calculators.py)
# ----------------------------------------------------------------------------------------------
def gold_contract():
    torch.manual_seed(700)
    full = ref_model.Generator(256, 512, 8)
    sd = full.state_dict()
    shape = Get_Network_Shape(sd)
    rm = Get_Uniform_RmveList(shape, 0.7)
    rng = np.random.RandomState(701)
    scores = [rng.rand(c) for c in shape]
    masks = Generate_Prune_Mask_List(scores, shape, rm)
    pruned = Mask_the_Generator(sd, masks)
    pshape = Get_Network_Shape(pruned)
    pg = Build_Generator_From_Dict(pruned, size=256)
    disc = ref_model.Discriminator(256)
    macs_full = Calculators.StyleGAN2_FLOPCal(sd)           # Util/Calculators.py:95-105
    macs_pruned = Calculators.StyleGAN2_FLOPCal(pg.state_dict())
    with open(os.path.join(OUT, "contract_256.json"), "w") as f:
        json.dump(dict(
            full_keys=[[k, list(v.shape)] for k, v in sd.items()],
            full_shape=[int(c) for c in shape],
            pruned_keys=[[k, list(v.shape)] for k, v in pg.state_dict().items()],
            pruned_shape=[int(c) for c in pshape],
            pruned_params=int(sum(p.numel() for p in pg.parameters())),
            full_params=int(sum(p.numel() for p in full.parameters())),
            d_keys=[[k, list(v.shape)] for k, v in disc.state_dict().items()],
            d_params=int(sum(p.numel() for p in disc.parameters())),
            macs_full=int(macs_full), macs_pruned=int(macs_pruned),
            kat_macs_256=int(Calculators.GENERATOR_FLOPS_256PX), kat_macs_1024=int(Calculators.GENERATOR_FLOPS_1024PX),
            n_latent=full.n_latent, num_layers=full.num_layers,
        ), f, indent=1)
    # a *tiny* prune-chain vector the product's own mask code can be checked against
    tiny = make_tiny_generator(702)
    tsd = tiny.state_dict()
    tshape = Get_Network_Shape(tsd)
    trm = Get_Uniform_RmveList(tshape, 0.5)
    tscores = [rng.rand(c) for c in tshape]
    tmasks = Generate_Prune_Mask_List(tscores, tshape, trm)
    tpruned = Mask_the_Generator(tsd, tmasks)
    arrs = sd_arrays("full/", tsd)
    arrs.update(sd_arrays("pruned/", tpruned))
    for i, (s, m) in enumerate(zip(tscores, tmasks)):
        arrs[f"score{i}"] = s
        arrs[f"mask{i}"] = np.asarray(m)
    arrs["rmve"] = np.asarray(trm)
    save("prune_chain_tiny", **arrs)



# ----------------------------------------------------------------------------------------------
# 8. one full training iteration (train.py:371-398): D step, R1, G+KD step, path-length reg, EMA
# ----------------------------------------------------------------------------------------------
def gold_train_iter():
    from Miscellaneous.distributed import reduce_sum, get_world_size
    student = make_tiny_generator(800, shape=[5, 5, 4, 4, 3, 3, 2, 2])
    g_ema = make_tiny_generator(800, shape=[5, 5, 4, 4, 3, 3, 2, 2])
    teacher = make_tiny_generator(801)
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    torch.manual_seed(802)
    disc = ref_model.Discriminator(TINY["size"])
    B = 4
    ns = dict(torch=torch, F=F, autograd=autograd, random=random, device="cpu", math=__import__("math"),
              Batch_Img_Parsing=Batch_Img_Parsing, Get_Masked_Tensor=Get_Masked_Tensor, reduce_sum=reduce_sum,
              get_world_size=get_world_size, train_hyperparams=types.SimpleNamespace(LPIPS_IMAGE_SIZE=256))
    lift_train_functions(["requires_grad", "KD_loss", "g_nonsaturating_loss", "d_logistic_loss", "d_r1_loss", "make_noise",
                          "mixing_noise", "index_aware_mixing_noise", "G_Loss_BackProp", "D_Loss_BackProp",
                          "D_Reg_BackProp", "G_Reg_BackProp"], ns)
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        cls[i][((yy - 250) / 190.0) ** 2 + ((xx - 260) / 160.0) ** 2 < 1.0] = 2 + i
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).float()
    args = types.SimpleNamespace(batch_size=B, latent=TINY["style_dim"], mixing=0.9, n_latent=student.n_latent,
                                 kd_mode="Output_Only", kd_l1_lambda=3, kd_lpips_lambda=3, size=TINY["size"], r1=10,
                                 d_reg_every=16, g_reg_every=4, path_regularize=2, path_batch_shrink=2)
    cg, cd = 4 / 5, 16 / 17
    g_optim = torch.optim.Adam(student.parameters(), lr=0.002 * cg, betas=(0.0 ** cg, 0.99 ** cg))
    d_optim = torch.optim.Adam(disc.parameters(), lr=0.002 * cd, betas=(0.0 ** cd, 0.99 ** cd))
    gen = torch.Generator().manual_seed(803)
    real_img = torch.rand(B, 3, TINY["size"], TINY["size"], generator=gen) * 2 - 1
    pl_noise = torch.randn(B // 2, 3, TINY["size"], TINY["size"], generator=gen)

    rec = RecordingNoise()
    calls = []
    orig_fwd = ref_model.Generator.forward

    def capturing_forward(self, noise_z, *a, **k):
        calls.append(([t.detach().clone() for t in noise_z], k.get("inject_index"), len(rec.log)))
        return orig_fwd(self, noise_z, *a, **k)

    out = {}
    out.update(sd_arrays("student_sd/", {k: v.clone() for k, v in student.state_dict().items()}))
    out.update(sd_arrays("teacher_sd/", teacher.state_dict()))
    out["real_img"], out["pl_noise"] = real_img, pl_noise
    d_sd0 = {k: v.clone() for k, v in disc.state_dict().items()}
    out["d_seed"] = np.int64(802)
    out["d_checksum"] = np.array([float(v.double().sum()) for v in d_sd0.values()])

    def chk(tensors):
        return np.array([[float(t.double().sum()), float(t.double().abs().sum())] for t in tensors])

    randints = []
    orig_randint = random.randint

    def logged_randint(a, b):
        v = orig_randint(a, b)
        randints.append(v)
        return v

    with mock.patch.object(ref_model.NoiseInjection, "forward", lambda self, image, noise=None: rec(self, image, noise)), \
            mock.patch.object(ref_model.Generator, "forward", capturing_forward), \
            mock.patch.object(random, "randint", logged_randint), \
            mock.patch.object(torch, "randn_like", lambda t: pl_noise):
        random.seed(900)
        torch.manual_seed(901)
        loss_dict = {}
        # --- D step (train.py:241-262)
        ns["D_Loss_BackProp"](student, disc, real_img, args, "cpu", loss_dict, d_optim)
        zs, inj, _ = calls[-1]
        nl = student.num_layers
        out["d/n_z"] = np.int64(len(zs))
        for i, z in enumerate(zs):
            out[f"d/z{i}"] = z
        out["d/inject_index"] = np.int64(-1 if inj is None else inj)   # mixing_noise: generator draws its own index
        for i in range(nl):
            out[f"d/noise{i}"] = rec.log[i]
        out["d/loss"], out["d/real_score"], out["d/fake_score"] = loss_dict["d"], loss_dict["real_score"], loss_dict["fake_score"]
        out["d/grad_chk"] = chk([p.grad for p in disc.parameters()])
        out["d/param_chk"] = chk([p.detach() for p in disc.parameters()])
        out["d/final_linear.1.weight.grad"] = disc.final_linear[1].weight.grad.clone()
        # --- R1 (train.py:264-278)
        r1 = ns["D_Reg_BackProp"](real_img, disc, args, d_optim)
        real_img.requires_grad = False
        out["r1/loss"] = r1
        out["r1/grad_chk"] = chk([p.grad for p in disc.parameters()])
        out["r1/param_chk"] = chk([p.detach() for p in disc.parameters()])
        out["r1/convs.0.1.bias.grad"] = disc.convs[0][1].bias.grad.clone()
        # --- G + KD step (train.py:280-308)
        n0 = len(rec.log)
        ns["G_Loss_BackProp"](student, disc, args, "cpu", loss_dict, g_optim, teacher, None, lambda x: (logits,))
        zs, inj, _ = calls[-2]
        out["g/n_z"] = np.int64(len(zs))
        for i, z in enumerate(zs):
            out[f"g/z{i}"] = z
        out["g/inject_index"] = np.int64(-1 if inj is None else inj)
        for i in range(nl):
            out[f"g/student_noise{i}"] = rec.log[n0 + i]
            out[f"g/teacher_noise{i}"] = rec.log[n0 + nl + i]
        out["g/g_loss"], out["g/kd_l1_loss"] = loss_dict["g"], loss_dict["kd_l1_loss"]
        for n, prm in student.named_parameters():
            out["g/grad/" + n] = prm.grad.detach().clone()
            out["g/param_after/" + n] = prm.detach().clone()
        # --- path-length regulariser (train.py:310-338)
        n1 = len(rec.log)
        path_loss, path_lengths, mean_pl, mean_pl_avg = ns["G_Reg_BackProp"](student, args, 0, g_optim)
        zs, inj, _ = calls[-1]
        out["pl/n_z"] = np.int64(len(zs))
        for i, z in enumerate(zs):
            out[f"pl/z{i}"] = z
        # with 2 latents and inject_index=None the forward draws random.randint itself (model.py:604-605)
        out["pl/inject_index"] = np.int64(randints[-1] if len(zs) > 1 else -1)
        out["pl/rand_state_seed"] = np.int64(900)
        for i in range(nl):
            out[f"pl/noise{i}"] = rec.log[n1 + i]
        out["pl/path_loss"], out["pl/path_lengths"], out["pl/mean_path_length"] = path_loss, path_lengths, mean_pl
        for n, prm in student.named_parameters():
            out["pl/grad/" + n] = prm.grad.detach().clone()
            out["pl/param_after/" + n] = prm.detach().clone()
    mask = ((cls > 0) * (cls != 16)).unsqueeze(0).float()
    mask = (F.interpolate(mask, scale_factor=TINY["size"] / 512, mode="bilinear", align_corners=False).squeeze() > 0.5).float()
    out["mask"] = mask.view(B, 1, TINY["size"], TINY["size"])
    # EMA (train.py:124-129, restated: the removed add_(Number, Tensor) overload == add_(tensor, alpha=number))
    accum = 0.5 ** (32 / (10 * 1000))
    pe, pg = dict(g_ema.named_parameters()), dict(student.named_parameters())
    with torch.no_grad():
        for k in pe:
            pe[k].mul_(accum).add_(pg[k], alpha=1 - accum)
            out["ema/" + k] = pe[k].detach().clone()
    # which inject index did the PL forward draw?  replay python's RNG: seeded 900, consumed in order by
    # D step (mixing_noise: random()), [generator forward: randint if 2 latents], G step (random(), randint), PL (random(), [randint])
    save("train_iter_tiny", **out)
    with open(os.path.join(OUT, "train_iter_tiny_meta.json"), "w") as f:
        json.dump(dict(batch=B, student_shape=[5, 5, 4, 4, 3, 3, 2, 2], teacher_shape=TINY["shape"],
                       lr_g=0.002 * cg, betas_g=[0.0 ** cg, 0.99 ** cg], lr_d=0.002 * cd, betas_d=[0.0 ** cd, 0.99 ** cd],
                       r1=10, d_reg_every=16, g_reg_every=4, path_regularize=2, path_batch_shrink=2, accum=accum,
                       calls=[dict(n_z=len(c[0]), inject_index=c[1]) for c in calls]), f, indent=1)



# ----------------------------------------------------------------------------------------------
# 9. content-aware saliency scores (Util/content_aware_pruning.py:152-196,200-249; prune.py:39-46)
# ----------------------------------------------------------------------------------------------
def gold_saliency():
    from Util.content_aware_pruning import Get_Salt_Pepper_Noisy_Image, Get_Weight_Gradient
    gnet = make_tiny_generator(950)
    B, size = 3, TINY["size"]
    rec = RecordingNoise()
    torch.manual_seed(951)
    np.random.seed(952)
    z = torch.randn(B, TINY["style_dim"])
    with mock.patch.object(ref_model.NoiseInjection, "forward", lambda self, image, noise=None: rec(self, image, noise)):
        img = gnet(noise_z=[z])                                   # randomize_noise=True, as :226
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    noisy_list, hits, sps = [], [], []
    for i in range(B):
        mask = ((yy - 15 - i) / 11.0) ** 2 + ((xx - 16 + i) / 9.0) ** 2 < 1.0     # stand-in for the BiSeNet face mask
        single = img[i:i + 1]
        noisy = Get_Salt_Pepper_Noisy_Image(single, mask, 0.3)     # the reference's own per-pixel loop (:152-171)
        noisy_list.append(noisy)
        hit = (noisy.detach() != single.detach()).any(1, keepdim=True)
        hits.append(hit.float())
        sps.append(torch.where(hit, noisy.detach()[:, :1], torch.zeros(())))
    grad_score = Get_Weight_Gradient(torch.cat(noisy_list), img, gnet)           # :174-196
    out = sd_arrays("sd/", gnet.state_dict())
    out["z"] = z
    for i, n in enumerate(rec.log):
        out[f"noise{i}"] = n
    out["hit"], out["sp"] = torch.cat(hits), torch.cat(sps)
    out["n_layers"] = np.int64(len(grad_score))
    for i, sc in enumerate(grad_score):
        out[f"score{i}"] = np.asarray(sc)
    save("saliency_tiny", **out)


# ----------------------------------------------------------------------------------------------
# 10. content mask: the reference's own Batch_Img_Parsing + Get_Masked_Tensor (Util/content_aware_pruning.py:61-117)
#     around a stand-in parsing net (BiSeNet's weights are not obtainable offline).  The logits are regenerated on the
#     test side from oracle/synth.py (integer hashing: machine independent); only their checksum is stored.
# ----------------------------------------------------------------------------------------------
def gold_content_mask():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import synth
    out = {}
    B = 2
    logits = torch.from_numpy(synth.parsing_logits(B, 512, 19, seed=1))
    out["logits_checksum"] = np.array([float(logits.double().sum()), float(logits.double().abs().sum())])
    for S in (32, 256, 1024):
        img = torch.from_numpy(synth.hash01((B, 3, S, S), seed=100 + S) * np.float32(2.6) - np.float32(1.3))   # exceeds [-1,1]: clamp matters
        seen = {}

        def net(x):
            seen["x"] = x.detach().clone()
            return (logits,)

        parsing = Batch_Img_Parsing(img, net, "cpu")                        # :61-88
        ones = torch.ones(B, 3, S, S)
        masked = Get_Masked_Tensor(ones, parsing, "cpu", mask_grad=False)    # :90-117  (img * mask, img = 1)
        mask = masked[:, 0]
        assert torch.equal(masked[:, 1], mask) and set(np.unique(mask.numpy())) <= {0.0, 1.0}
        out[f"S{S}/parse_in_sub"] = seen["x"][:, :, ::7, ::5]
        out[f"S{S}/parse_in_sum"] = np.array([float(seen["x"].double().sum()), float(seen["x"].double().abs().sum())])
        out[f"S{S}/mask_bits"] = np.packbits(mask.numpy().astype(np.uint8).reshape(-1))
        out[f"S{S}/mask_sum"] = np.int64(mask.sum().item())
    out["parsing_sum"] = np.array([int(parsing.sum()), int((parsing * torch.arange(512)[None, None, :]).sum())], dtype=np.int64)
    # the batch-1 case Get_Masked_Tensor cannot run (its .squeeze() drops the batch axis, App. D-5) is covered by
    # batch independence on the test side
    save("content_mask", **out)


# ----------------------------------------------------------------------------------------------
# 11. KD loss variants: kd_mode='Intermediate' (train.py:165-169) and the LPIPS term's data flow (train.py:173-182) with a
#     stand-in perceptual distance (LPIPS weights are not obtainable offline) — same models as gold_kd_step
# ----------------------------------------------------------------------------------------------
def gold_kd_modes():
    B = 4
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        r = ((yy - 256 - 10 * i) / 200.0) ** 2 + ((xx - 256 + 7 * i) / 150.0) ** 2
        cls[i][r < 1.0] = 1 + i
        cls[i][(yy > 440)] = 16
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).float()
    percept = lambda a, b: ((a - b) ** 2).mean(dim=[1, 2, 3])     # stand-in for lpips.PerceptualLoss: [B] distances
    out = {}
    for mode, with_percept in (("Intermediate", False), ("Intermediate", True), ("Output_Only", True)):
        student = make_tiny_generator(300, shape=[5, 5, 4, 4, 3, 3, 2, 2])
        teacher = make_tiny_generator(301)
        teacher.eval()
        for p in teacher.parameters():
            p.requires_grad = False
        torch.manual_seed(302)
        disc = ref_model.Discriminator(TINY["size"])
        ns = dict(torch=torch, F=F, autograd=autograd, random=random, device="cpu",
                  Batch_Img_Parsing=Batch_Img_Parsing, Get_Masked_Tensor=Get_Masked_Tensor,
                  train_hyperparams=types.SimpleNamespace(LPIPS_IMAGE_SIZE=256))
        lift_train_functions(["requires_grad", "KD_loss", "g_nonsaturating_loss", "make_noise",
                              "index_aware_mixing_noise", "G_Loss_BackProp", "Downsample_Image_256"], ns)
        args = types.SimpleNamespace(batch_size=B, latent=TINY["style_dim"], mixing=0.9, n_latent=student.n_latent,
                                     kd_mode=mode, kd_l1_lambda=3, kd_lpips_lambda=3, size=TINY["size"])
        lr, c = 0.002, 4 / 5
        g_optim = torch.optim.Adam(student.parameters(), lr=lr * c, betas=(0.0 ** c, 0.99 ** c))
        rec = RecordingNoise()
        captured = {}
        orig_fwd = ref_model.Generator.forward

        def capturing_forward(self, noise_z, *a, **k):
            captured.setdefault("calls", []).append(([t.detach().clone() for t in noise_z], k.get("inject_index")))
            return orig_fwd(self, noise_z, *a, **k)

        tag = f"{mode}{'_percept' if with_percept else ''}/"
        with mock.patch.object(ref_model.NoiseInjection, "forward", lambda self, image, noise=None: rec(self, image, noise)), \
                mock.patch.object(ref_model.Generator, "forward", capturing_forward):
            random.seed(400)
            torch.manual_seed(500)
            loss_dict = {}
            ns["G_Loss_BackProp"](student, disc, args, "cpu", loss_dict, g_optim, teacher, percept if with_percept else None,
                                  lambda x: (logits,))
        (zs, inj), _ = captured["calls"]
        nl = student.num_layers
        out[tag + "n_z"] = np.int64(len(zs))
        for i, z in enumerate(zs):
            out[tag + f"z{i}"] = z
        out[tag + "inject_index"] = np.int64(-1 if inj is None else inj)
        for i in range(nl):
            out[tag + f"student_noise{i}"] = rec.log[i]
            out[tag + f"teacher_noise{i}"] = rec.log[nl + i]
        for k in ("g", "kd_l1_loss", "kd_lpips_loss"):
            out[tag + k] = loss_dict[k]
        for n, prm in student.named_parameters():
            out[tag + "grad/" + n] = prm.grad.detach().clone()
    out["d_seed"] = np.int64(302)
    save("kd_modes_tiny", **out)


# ----------------------------------------------------------------------------------------------
# 12. float64 evaluations of the REFERENCE on the same inputs as the fp32 goldens above (discriminator32, generator_tiny
#     case (b), kd_step_tiny step 0).  They settle whether a deviation of the HIP path from an fp32 golden is the HIP
#     path's error or the fp32 reference's own rounding (LeakyReLU gate flips on random nets): tests assert
#     HIP-vs-float64 <= max(1e-3, 3 x fp32-reference-vs-float64), the latter number stored here.
# ----------------------------------------------------------------------------------------------
def _rel(a, b):
    den = b.abs().max().item()
    return (a.double() - b).abs().max().item() / (den if den > 0 else 1.0)


def gold_float64():
    out = {}
    # --- Discriminator(32): same seeds as gold_discriminator
    torch.manual_seed(600)
    disc = ref_model.Discriminator(TINY["size"])
    g = torch.Generator().manual_seed(601)
    x = torch.randn(4, 3, TINY["size"], TINY["size"], generator=g)
    x32 = x.clone().requires_grad_(True)
    y32 = disc(x32)
    (gx32,) = autograd.grad(F.softplus(-y32).mean(), x32)
    disc64 = ref_model.Discriminator(TINY["size"]).double()
    disc64.load_state_dict({k: v.double() for k, v in disc.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    y64 = disc64(x64)
    (gx64,) = autograd.grad(F.softplus(-y64).mean(), x64)
    out["d32/y"], out["d32/gx"] = y64, gx64
    out["d32/fp32_err_y"], out["d32/fp32_err_gx"] = np.float64(_rel(y32, y64)), np.float64(_rel(gx32, gx64))
    # --- tiny generator case (b): every parameter gradient of |img|.mean()
    gnet = make_tiny_generator(200)
    gtor = torch.Generator().manual_seed(201)
    z0 = torch.randn(3, TINY["style_dim"], generator=gtor)
    img32 = gnet([z0], randomize_noise=False)
    gnet.zero_grad()
    img32.abs().mean().backward()
    g32 = {n: p.grad.clone() for n, p in gnet.named_parameters()}
    gnet64 = make_tiny_generator(200).double()
    gnet64.load_state_dict({k: v.double() for k, v in gnet.state_dict().items()})
    img64 = gnet64([z0.double()], randomize_noise=False)
    img64.abs().mean().backward()
    out["gen/img"] = img64
    out["gen/fp32_err_img"] = np.float64(_rel(img32, img64))
    for n, p in gnet64.named_parameters():
        out["gen/grad/" + n] = p.grad.clone()
        out["gen/fp32_err/" + n] = np.float64(_rel(g32[n], p.grad))
    # --- KD step 0 of gold_kd_step, replayed in float64 through the reference's own G_Loss_BackProp
    with np.load(os.path.join(OUT, "kd_step_tiny.npz")) as z:
        kd = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    B = 4
    student = make_tiny_generator(300, shape=[5, 5, 4, 4, 3, 3, 2, 2]).double()
    teacher = make_tiny_generator(301).double()
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    for k, v in student.state_dict().items():
        assert torch.equal(v.float(), kd["student_sd/" + k].float()), k
    torch.manual_seed(302)
    disc = ref_model.Discriminator(TINY["size"]).double()   # randn in float64 differs from the fp32 draw: copy the fp32 weights
    torch.manual_seed(302)
    d32 = ref_model.Discriminator(TINY["size"])
    disc.load_state_dict({k: v.double() for k, v in d32.state_dict().items()})
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        r = ((yy - 256 - 10 * i) / 200.0) ** 2 + ((xx - 256 + 7 * i) / 150.0) ** 2
        cls[i][r < 1.0] = 1 + i
        cls[i][(yy > 440)] = 16
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).double()
    ns = dict(torch=torch, F=F, autograd=autograd, random=random, device="cpu",
              Batch_Img_Parsing=Batch_Img_Parsing, Get_Masked_Tensor=_get_masked_tensor_any_dtype,
              train_hyperparams=types.SimpleNamespace(LPIPS_IMAGE_SIZE=256))
    lift_train_functions(["requires_grad", "KD_loss", "g_nonsaturating_loss", "make_noise",
                          "index_aware_mixing_noise", "G_Loss_BackProp"], ns)
    n_z = int(kd["step0/n_z"])
    zs = [kd[f"step0/z{i}"].double() for i in range(n_z)]
    inj = int(kd["step0/inject_index"])
    ns["index_aware_mixing_noise"] = lambda *a, **k: (zs, None if inj < 0 else inj)     # replay the recorded draw
    nl = student.num_layers
    replay = [kd[f"step0/student_noise{i}"].double() for i in range(nl)] + [kd[f"step0/teacher_noise{i}"].double() for i in range(nl)]
    it = iter(replay)
    args = types.SimpleNamespace(batch_size=B, latent=TINY["style_dim"], mixing=0.9, n_latent=student.n_latent,
                                 kd_mode="Output_Only", kd_l1_lambda=3, kd_lpips_lambda=3, size=TINY["size"])
    c = 4 / 5
    g_optim = torch.optim.SGD(student.parameters(), lr=0.0)     # gradients only: the update is not part of this fixture
    loss_dict = {}
    with mock.patch.object(ref_model.NoiseInjection, "forward", lambda self, image, noise=None: image + self.weight * next(it)):
        ns["G_Loss_BackProp"](student, disc, args, "cpu", loss_dict, g_optim, teacher, None, lambda x: (logits,))
    out["kd/g_loss"], out["kd/kd_l1_loss"] = loss_dict["g"], loss_dict["kd_l1_loss"]
    out["kd/fp32_err_g_loss"] = np.float64(abs(float(kd["step0/g_loss"]) - loss_dict["g"].item()))
    for n, prm in student.named_parameters():
        out["kd/grad/" + n] = prm.grad.detach().clone()
        out["kd/fp32_err/" + n] = np.float64(_rel(kd["step0/grad/" + n], prm.grad))
    save("float64_refs", **out)
    worst = max(float(v) for k, v in out.items() if "fp32_err" in k)
    print("worst fp32-reference vs float64 deviation:", worst, "| D input grad:", float(out["d32/fp32_err_gx"]))


def _get_masked_tensor_any_dtype(img_tensor, batch_parsing, device, mask_grad=False):
    """Get_Masked_Tensor hard-codes torch.FloatTensor (Util/content_aware_pruning.py:104,107), which cannot multiply a
    float64 image in place of `masked_img_tensor[i] = ...`'s dtype; same arithmetic, image dtype."""
    masked = Get_Masked_Tensor(torch.ones_like(img_tensor, dtype=torch.float32), batch_parsing, device, mask_grad=False)
    return img_tensor * masked.to(img_tensor.dtype)


if __name__ == "__main__":
    # float64 / kd_modes READ kd_step_tiny.npz and the other fixtures: they come last so that a from-scratch regeneration works in one go
    which = sys.argv[1:] or ["fused_act", "upfirdn2d", "modconv", "generator", "generator512", "kd_step", "discriminator", "contract", "train_iter", "saliency", "content_mask", "float64", "kd_modes"]
    for w in which:
        print("==", w)
        globals()["gold_" + w]()
    leaked = [f for f in os.listdir(os.path.join(REF, "op")) if f.endswith(".hip")]
    assert not leaked, f"hipify wrote into the reference tree: {leaked}"
