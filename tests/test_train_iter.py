"""the rest of the training iteration (D step, R1, G+KD step, path-length regulariser, EMA) against
a golden captured from the reference's own D_Loss_BackProp / D_Reg_BackProp / G_Loss_BackProp / G_Reg_BackProp
(lifted from train.py by oracle/gen_golden.py).  The 29 MB discriminator is regenerated from its seed; its gradients
and updated weights are pinned by per-tensor (sum, abs-sum) checksums plus two small tensors in full."""
from unittest import mock

import numpy as np
import pytest
import torch

import cagc.model as M
from cagc import kd
from oracle import ref_kd
from oracle.ref_model import regenerate_state_dict
from _util import assert_close, load_json, load_npz, sub


def _objects(g, meta, dev):
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(g, "student_sd/"), strict=True)
    ema = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    ema.load_state_dict(sub(g, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
    disc = M.Discriminator(32)
    sd = regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"])
    chk = torch.tensor([float(v.double().sum()) for v in sd.values()], dtype=torch.float64)
    assert torch.allclose(chk, g["d_checksum"].double(), rtol=0, atol=1e-6)
    disc.load_state_dict(sd, strict=True)
    return student.to(dev), teacher.to(dev), disc.to(dev), ema.to(dev)


def _check_chk(tensors, gold, tol, what):
    got = torch.tensor([[float(t.double().sum()), float(t.double().abs().sum())] for t in tensors], dtype=torch.float64)
    gold = gold.double()
    scale = gold[:, 1].clamp_min(1e-12)   # compare both sums relative to the abs-sum of the tensor
    err = ((got - gold).abs() / scale[:, None]).max().item()
    assert err <= tol, f"{what}: checksum rel err {err:.3e} > {tol:.1e}"


def _run(dev, tol, tol_chk):
    g = load_npz("train_iter_tiny")
    meta = load_json("train_iter_tiny_meta")
    student, teacher, disc, ema = _objects(g, meta, dev)
    it = kd.TrainIteration(student, teacher, disc, g_ema=ema, latent=24)
    assert abs(it.d_optim.param_groups[0]["lr"] - meta["lr_d"]) < 1e-12 and np.allclose(it.d_optim.param_groups[0]["betas"], meta["betas_d"])
    to = lambda t: t.to(dev)
    nl = student.num_layers
    # D step
    zs = [to(g[f"d/z{i}"]) for i in range(g["d/n_z"])]
    out = it.d_step(to(g["real_img"]), zs, None if g["d/inject_index"] < 0 else g["d/inject_index"],
                    noise=[to(g[f"d/noise{i}"]) for i in range(nl)])
    assert abs(out["d"].item() - float(g["d/loss"])) < tol * max(1, abs(float(g["d/loss"])))
    assert abs(out["real_score"].item() - float(g["d/real_score"])) < 10 * tol and abs(out["fake_score"].item() - float(g["d/fake_score"])) < 10 * tol
    # gradients were consumed by the optimiser step; the updated weights pin them (Adam's first step = -lr*sign(g))
    _check_chk([p.detach() for p in disc.parameters()], g["d/param_chk"], tol_chk, "D params after D step")
    # R1
    r1 = it.d_reg(to(g["real_img"]))
    assert abs(r1.item() - float(g["r1/loss"])) < 5 * tol * max(1, abs(float(g["r1/loss"])))
    assert_close(disc.convs[0][1].bias.grad, g["r1/convs.0.1.bias.grad"], 20 * tol, "R1 grad convs.0.1.bias")
    _check_chk([p.detach() for p in disc.parameters()], g["r1/param_chk"], tol_chk, "D params after R1")
    # G + KD step
    zs = [to(g[f"g/z{i}"]) for i in range(g["g/n_z"])]
    losses = it.g_step(zs, None if g["g/inject_index"] < 0 else g["g/inject_index"], to(g["mask"]),
                       student_noise=[to(g[f"g/student_noise{i}"]) for i in range(nl)],
                       teacher_noise=[to(g[f"g/teacher_noise{i}"]) for i in range(nl)])
    assert abs(losses["g"].item() - float(g["g/g_loss"])) < 10 * tol * max(1, abs(float(g["g/g_loss"])))
    assert abs(losses["kd_l1_loss"].item() - float(g["g/kd_l1_loss"])) < 10 * tol
    params = dict(student.named_parameters())
    for k, v in sub(g, "g/grad/").items():
        assert_close(params[k].grad, v, 30 * tol if v.numel() > 1 else 300 * tol, "G step grad " + k)
    with torch.no_grad():
        for k, v in sub(g, "g/param_after/").items():
            params[k].copy_(to(v))
    # path-length regulariser (second order)
    zs = [to(g[f"pl/z{i}"]) for i in range(g["pl/n_z"])]
    with mock.patch.object(torch, "randn_like", lambda t: to(g["pl_noise"])):
        path_loss, pl = it.g_reg(zs, None if g["pl/inject_index"] < 0 else g["pl/inject_index"],
                                 noise=[to(g[f"pl/noise{i}"]) for i in range(nl)])
    assert_close(pl, g["pl/path_lengths"], 10 * tol, "path lengths")
    assert abs(path_loss.item() - float(g["pl/path_loss"])) < 10 * tol
    for k, v in sub(g, "pl/grad/").items():
        assert_close(params[k].grad, v, 100 * tol if v.numel() > 1 else 1000 * tol, "PL grad " + k)
    with torch.no_grad():
        for k, v in sub(g, "pl/param_after/").items():
            params[k].copy_(to(v))
    it.ema()
    pe = dict(ema.named_parameters())
    for k, v in sub(g, "ema/").items():
        assert_close(pe[k].detach(), v, 1e-6, "EMA " + k)


def test_full_iteration_cpu_matches_reference():
    _run("cpu", 2e-5, 2e-5)


@pytest.mark.gpu
def test_full_iteration_gpu_matches_reference():
    """The golden on the GPU with the F(2x2) Winograd kernel on every stride-1 3x3 layer (CAGC_WINO_F4=0, read once per process:
    own interpreter) at the bounds the golden has always been held to."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import test_train_iter as t; t._run('cuda', 1e-4, 2e-4); print('ITER_OK')"
            % (root, os.path.join(root, "content-aware-gan-compression_amd"), os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CAGC_WINO_F4="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ITER_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_full_iteration_gpu_matches_reference_default_kernels():
    """Same golden with the library's DEFAULT launch policy, at the north-star bar: every G-step gradient <= 30 * 3.3e-5 = 1e-3 of the
    reference's fp32 values with NO gate forcing (observed: passes at 3.4e-5).  Since round 4 F(4x4)
    Winograd is chosen per launch, only when the grid fills the chip — D(32)'s 512-channel 32x32 layers at batch 4 are 128 workgroups and
    take their F(2x2) packing, so the default kernels meet the bounds the golden has always been held to."""
    _run("cuda", 3.3e-5, 2e-4)


@pytest.mark.gpu
def test_full_iteration_gpu_trajectory_with_f4_forced():
    """F(4x4,3x3) forced on every eligible layer (`wino4_min_wgs` = 0): its activations differ from the fp32 reference's by ~1e-5 instead
    of ~1e-6, so more LeakyReLU gates of the tiny random discriminator land on the other side of 0, each moving a 3x3 patch of a gradient
    (DESIGN §2); observed up to 9.1e-3 on a student gradient of this 4-sample, 32 px golden — this run only pins the trajectory loosely.
    The kernel itself is held to 5e-5 per layer (tests/test_wino4_gpu.py), the real-size step to the common-gate protocol AND to 1e-3 on
    the oracle's own gates (tests/test_bench_selection_gpu.py)."""
    from cagc import _lib
    with _lib.tuning(wino4_min_wgs=0):
        _run("cuda", 5e-4, 5e-4)


def test_oracle_full_iteration_pieces_match_reference():
    g = load_npz("train_iter_tiny")
    d_sd = regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"])
    student = sub(g, "student_sd/")
    nl = 7
    zs = [g[f"d/z{i}"] for i in range(g["d/n_z"])]
    d_loss, rs, fs = ref_kd.d_losses_ref(student, d_sd, g["real_img"], zs, None, [g[f"d/noise{i}"] for i in range(nl)])
    assert abs(d_loss.item() - float(g["d/loss"])) < 2e-5 and abs(rs.item() - float(g["d/real_score"])) < 2e-4
    # path-length pieces on the post-G-step student
    post = dict(student)
    post.update(sub(g, "g/param_after/"))
    post = {k: (v.clone().requires_grad_(True) if k in sub(g, "g/param_after/") else v) for k, v in post.items()}
    zs = [g[f"pl/z{i}"] for i in range(g["pl/n_z"])]
    pl_loss, pl, mean, _ = ref_kd.path_reg_ref(post, zs, int(g["pl/inject_index"]), [g[f"pl/noise{i}"] for i in range(nl)], g["pl_noise"])
    assert_close(pl, g["pl/path_lengths"], 1e-4, "oracle path lengths")
    assert abs(pl_loss.item() - float(g["pl/path_loss"])) < 1e-5 and abs(mean.item() - float(g["pl/mean_path_length"])) < 1e-6


# ---------------------------------------------------------------------------------------------------
# DDP on D: D step + R1 on 2 gloo ranks == 1 rank on the concatenated batch.  Rank r gets samples
# r::2, which keeps D's minibatch-stddev groups identical (model.py:784-790 groups sample j with j + B/4 ...).
# ---------------------------------------------------------------------------------------------------
def _dd_inputs():
    gen = torch.Generator().manual_seed(77)
    B = 8
    real = torch.rand(B, 3, 32, 32, generator=gen) * 2 - 1
    zs = [torch.randn(B, 24, generator=gen), torch.randn(B, 24, generator=gen)]
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(7)]
    return real, zs, noise


def _dd_worker(rank, world, port, tmp):
    import os
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cagc import distributed as cd
    torch.set_num_threads(2)
    cd.init_from_env(backend="gloo")
    g = load_npz("train_iter_tiny")
    meta = load_json("train_iter_tiny_meta")
    student, teacher, disc, ema = _objects(g, meta, "cpu")
    it = kd.TrainIteration(student, teacher, cd.wrap_discriminator(disc, torch.device("cpu")), g_ema=ema, latent=24)
    real, zs, noise = _dd_inputs()
    sl = slice(rank, None, world)
    # R1 first, on the identical initial weights: its gradients are then comparable at rounding level (after an optimiser
    # step the replicas' weights differ in the last bits, which flips LeakyReLU gates of this random net)
    r1 = it.d_reg(real[sl])
    r1 = cd.reduce_loss_dict({"r1": r1})["r1"]
    snap_g = {k: v.grad.detach().clone() for k, v in disc.named_parameters()}      # R1 gradients, averaged by DDP
    it.d_step(real[sl], [z[sl] for z in zs], 3, noise=[n[sl] for n in noise])
    snap = {k: v.detach().clone() for k, v in disc.named_parameters()}
    snap_g2 = {k: v.grad.detach().clone() for k, v in disc.named_parameters()}
    # the frozen-D generator step must bypass the DDP wrapper (no gradient reaches D's parameters there)
    it.g_step([z[sl] for z in zs], 3, kd.ellipse_mask(4, 32, "cpu"), student_noise=[n[sl] for n in noise], teacher_noise=[n[sl] for n in noise])
    it.d_step(real[sl], [z[sl] for z in zs], 3, noise=[n[sl] for n in noise])      # and DDP still works afterwards
    if rank == 0:
        torch.save({"d": snap, "g": snap_g, "g2": snap_g2, "r1": r1.item()}, tmp)
    cd.barrier()
    dist.destroy_process_group()


def test_ddp_discriminator_two_ranks_equal_one_rank(tmp_path):
    import os
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "dd.pt")
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_dd_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    g = load_npz("train_iter_tiny")
    meta = load_json("train_iter_tiny_meta")
    student, teacher, disc, ema = _objects(g, meta, "cpu")
    it = kd.TrainIteration(student, teacher, disc, g_ema=ema, latent=24)
    real, zs, noise = _dd_inputs()
    r1 = it.d_reg(real)
    assert abs(r1.item() - got["r1"]) < 1e-5 * max(1.0, abs(r1.item()))
    gmax = max(float(v.grad.abs().max()) for v in disc.parameters())
    for k, v in disc.named_parameters():
        # bias gradients of R1 exist only through the minibatch-stddev channel (1e-5 of the weight gradients): absolute bound
        err = (got["g"][k] - v.grad).abs().max().item()
        assert err <= 1e-4 * max(float(v.grad.abs().max()), 1e-3 * gmax), f"DDP R1 grad {k}: {err:.3e}"
    it.d_step(real, zs, 3, noise=noise)
    for k, v in disc.named_parameters():
        assert_close(got["g2"][k], v.grad, 5e-3, "DDP D-step grad " + k)   # after one Adam step: last-bit weight differences flip gates
        # two Adam steps with beta1 = 0 (update = lr * g / sqrt(v)) amplify rounding-level gradient differences
        assert_close(got["d"][k], v.detach(), 2e-3, "DDP D param " + k)
