"""north_star: "prune.py and train.py drop in unchanged".  The reference's OWN step functions — D_Loss_BackProp,
D_Reg_BackProp, G_Loss_BackProp (+ KD_loss, g_nonsaturating_loss, index_aware_mixing_noise), G_Reg_BackProp
(train.py:145-184, 187-237, 241-338) — are lifted out of train.py's AST at test time (the file cannot be imported:
argparse and `cuda:0` at module scope) and executed UNMODIFIED with `model` / `op` resolving to this package, on the
product's Generator / Discriminator.  With the same seeds as oracle/gen_golden.py `gold_train_iter`, the run must land on
the `train_iter_tiny` golden, which the same functions produced on the reference's own model.

Runs only where the reference checkout is mounted (the build container); nothing of the reference is copied."""
import ast
import os
import random
import sys
import types
from unittest import mock

import numpy as np
import pytest
import torch
from torch import autograd
from torch.nn import functional as F

from _util import assert_close, load_json, load_npz, sub
from oracle.ref_model import regenerate_state_dict

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference checkout not mounted")

STEP_FUNCTIONS = ["requires_grad", "KD_loss", "g_nonsaturating_loss", "d_logistic_loss", "d_r1_loss", "make_noise", "mixing_noise",
                  "index_aware_mixing_noise", "G_Loss_BackProp", "D_Loss_BackProp", "D_Reg_BackProp", "G_Reg_BackProp"]


@pytest.fixture()
def reference_namespace():
    """The lifted step functions in a namespace whose helpers are the reference's own (Util.content_aware_pruning,
    Miscellaneous.distributed) and whose `model` is the PRODUCT."""
    import model as product_model    # tests/conftest.py puts content-aware-gan-compression_amd first on sys.path
    assert "content-aware-gan-compression_amd" in product_model.__file__
    stubbed = ("torchvision", "torchvision.utils", "torchvision.transforms", "PIL", "PIL.Image")
    saved = {k: sys.modules.get(k) for k in stubbed}
    for name in stubbed:                      # imported at module scope by the reference's helpers, never used by these functions
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    sys.path.append(REF)                      # AFTER the product: `model` / `op` stay the product's
    old_bytecode = sys.dont_write_bytecode
    sys.dont_write_bytecode = True            # nothing is written into the read-only checkout
    try:
        from Util import network_util
        from Util.content_aware_pruning import Batch_Img_Parsing, Get_Masked_Tensor
        from Miscellaneous.distributed import get_world_size, reduce_sum
        assert network_util.Generator is product_model.Generator
        with open(os.path.join(REF, "train.py")) as f:
            tree = ast.parse(f.read())
        picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in STEP_FUNCTIONS]
        assert {n.name for n in picked} == set(STEP_FUNCTIONS), "train.py layout changed"
        ns = dict(torch=torch, F=F, autograd=autograd, random=random, device="cpu", math=__import__("math"),
                  Batch_Img_Parsing=Batch_Img_Parsing, Get_Masked_Tensor=Get_Masked_Tensor, reduce_sum=reduce_sum,
                  get_world_size=get_world_size, train_hyperparams=types.SimpleNamespace(LPIPS_IMAGE_SIZE=256))
        exec(compile(ast.Module(body=picked, type_ignores=[]), "<lifted from reference train.py>", "exec"), ns)
        yield ns, product_model
    finally:
        sys.dont_write_bytecode = old_bytecode
        sys.path.remove(REF)
        for k in [m for m in sys.modules if m.split(".")[0] in ("Util", "Miscellaneous")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _chk(tensors):
    return torch.tensor([[float(t.double().sum()), float(t.double().abs().sum())] for t in tensors], dtype=torch.float64)


def _chk_close(tensors, gold, tol, what):
    got, gold = _chk(tensors), gold.double()
    err = ((got - gold).abs() / gold[:, 1].clamp_min(1e-12)[:, None]).max().item()
    assert err <= tol, f"{what}: checksum rel err {err:.3e} > {tol:.1e}"


def test_reference_train_step_functions_run_unchanged_on_the_product(reference_namespace):
    ns, M = reference_namespace
    g, meta = load_npz("train_iter_tiny"), load_json("train_iter_tiny_meta")
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(g, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    disc = M.Discriminator(32)
    disc.load_state_dict(regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
    B, size = meta["batch"], 32
    # the stand-in parsing net of the golden: class maps from a closed formula (BiSeNet's weights are not obtainable offline)
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        cls[i][((yy - 250) / 190.0) ** 2 + ((xx - 260) / 160.0) ** 2 < 1.0] = 2 + i
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).float()
    args = types.SimpleNamespace(batch_size=B, latent=24, mixing=0.9, n_latent=student.n_latent, kd_mode="Output_Only",
                                 kd_l1_lambda=3, kd_lpips_lambda=3, size=size, r1=meta["r1"], d_reg_every=meta["d_reg_every"],
                                 g_reg_every=meta["g_reg_every"], path_regularize=meta["path_regularize"],
                                 path_batch_shrink=meta["path_batch_shrink"])
    g_optim = torch.optim.Adam(student.parameters(), lr=meta["lr_g"], betas=tuple(meta["betas_g"]))
    d_optim = torch.optim.Adam(disc.parameters(), lr=meta["lr_d"], betas=tuple(meta["betas_d"]))
    real_img = g["real_img"].clone()
    tol = 2e-5      # the bar of test_train_iter.py's CPU run of the product's own TrainIteration
    with mock.patch.object(torch, "randn_like", lambda t: g["pl_noise"]):
        random.seed(900)          # the golden's seeds: the product draws latents / noise maps / mixing indices in the reference's order
        torch.manual_seed(901)
        loss_dict = {}
        ns["D_Loss_BackProp"](student, disc, real_img, args, "cpu", loss_dict, d_optim)                       # train.py:241-262
        assert abs(float(loss_dict["d"]) - float(g["d/loss"])) < tol * max(1, abs(float(g["d/loss"])))
        assert abs(float(loss_dict["real_score"]) - float(g["d/real_score"])) < 10 * tol
        assert abs(float(loss_dict["fake_score"]) - float(g["d/fake_score"])) < 10 * tol
        assert_close(disc.final_linear[1].weight.grad, g["d/final_linear.1.weight.grad"], 20 * tol, "D step grad final_linear.1.weight")
        _chk_close([p.detach() for p in disc.parameters()], g["d/param_chk"], tol, "D params after D step")
        r1 = ns["D_Reg_BackProp"](real_img, disc, args, d_optim)                                                # train.py:264-278
        real_img.requires_grad = False
        assert abs(float(r1) - float(g["r1/loss"])) < 5 * tol * max(1, abs(float(g["r1/loss"])))
        assert_close(disc.convs[0][1].bias.grad, g["r1/convs.0.1.bias.grad"], 20 * tol, "R1 grad convs.0.1.bias")
        _chk_close([p.detach() for p in disc.parameters()], g["r1/param_chk"], tol, "D params after R1")
        ns["G_Loss_BackProp"](student, disc, args, "cpu", loss_dict, g_optim, teacher, None, lambda x: (logits,))  # train.py:280-308
        assert abs(float(loss_dict["g"]) - float(g["g/g_loss"])) < 10 * tol * max(1, abs(float(g["g/g_loss"])))
        assert abs(float(loss_dict["kd_l1_loss"]) - float(g["g/kd_l1_loss"])) < 10 * tol
        params = dict(student.named_parameters())
        for k, v in sub(g, "g/grad/").items():
            assert_close(params[k].grad, v, 30 * tol if v.numel() > 1 else 300 * tol, "G step grad " + k)
        for k, v in sub(g, "g/param_after/").items():
            assert_close(params[k].detach(), v, 1e-4, "G step param " + k)
        with torch.no_grad():       # follow the golden's trajectory exactly into the regulariser
            for k, v in sub(g, "g/param_after/").items():
                params[k].copy_(v)
        path_loss, path_lengths, mean_pl, _ = ns["G_Reg_BackProp"](student, args, 0, g_optim)                   # train.py:310-338
        assert_close(path_lengths, g["pl/path_lengths"], 10 * tol, "path lengths")
        assert abs(float(path_loss) - float(g["pl/path_loss"])) < 10 * tol
        assert abs(float(mean_pl) - float(g["pl/mean_path_length"])) < 10 * tol
        for k, v in sub(g, "pl/grad/").items():
            assert_close(params[k].grad, v, 100 * tol if v.numel() > 1 else 1000 * tol, "PL grad " + k)
    assert all(q.grad is None for q in teacher.parameters())
