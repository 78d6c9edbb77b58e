"""Host logic of the KD step on CPU: loss definition, gradient flow, Adam hyper-parameters against the golden
captured from the reference's own G_Loss_BackProp; prune-chain surgery against the reference's Mask_the_Generator;
2-rank gloo DDP == 1-rank on the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cagc.model as M
from cagc import kd, prune
from oracle.ref_model import regenerate_state_dict
from _util import assert_close, load_json, load_npz, sub


def _kd_objects(g, meta):
    student = M.Generator(32, 24, 2, generator_net_shape=meta["student_shape"])
    student.load_state_dict(sub(g, "student_sd/"), strict=True)
    teacher = M.Generator(32, 24, 2, generator_net_shape=meta["teacher_shape"])
    teacher.load_state_dict(sub(g, "teacher_sd/"), strict=True)
    disc = M.Discriminator(32)
    disc.load_state_dict(regenerate_state_dict(load_json("discriminator32_keys"), g["d_seed"]), strict=True)
    return student, teacher, disc


def test_kd_step_matches_reference_g_loss_backprop():
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _kd_objects(g, meta)
    step = kd.KDStep(student, teacher, disc, latent=24)
    assert abs(step.optim.param_groups[0]["lr"] - meta["lr"]) < 1e-12
    assert np.allclose(step.optim.param_groups[0]["betas"], meta["betas"])
    for st in meta["steps"]:
        p = f"step{st['step']}/"
        nl = student.num_layers
        zs = [g[p + f"z{i}"] for i in range(st["n_z"])]
        losses = step.g_step(zs, st["inject_index"], g["mask"],
                             student_noise=[g[p + f"student_noise{i}"] for i in range(nl)],
                             teacher_noise=[g[p + f"teacher_noise{i}"] for i in range(nl)])
        assert abs(losses["g"].item() - float(g[p + "g_loss"])) < 3e-5
        assert abs(losses["kd_l1_loss"].item() - float(g[p + "kd_l1_loss"])) < 3e-5
        params = dict(student.named_parameters())
        for k, v in sub(g, p + "grad/").items():
            # single-element grads (noise.weight) are heavily cancelling sums: looser relative bound
            assert_close(params[k].grad, v, 3e-4 if v.numel() > 1 else 3e-3, f"step{st['step']} grad {k}")
        for k, v in sub(g, p + "param_after/").items():
            assert_close(params[k].detach(), v, 1e-4, f"step{st['step']} param {k}")
        with torch.no_grad():   # follow the reference trajectory exactly for the next step
            for k, v in sub(g, p + "param_after/").items():
                params[k].copy_(v)
        assert all(q.grad is None for q in disc.parameters()) and all(q.grad is None for q in teacher.parameters())


def test_prune_chain_matches_reference_mask_the_generator():
    g = load_npz("prune_chain_tiny")
    full, pruned = sub(g, "full/"), sub(g, "pruned/")
    shape = prune.network_shape(full)
    n = len(shape)
    scores = [g[f"score{i}"].numpy() for i in range(n)]
    rm = prune.uniform_remove_list(shape, 0.5)
    assert rm == [int(v) for v in g["rmve"]]
    masks = prune.masks_from_scores(scores, shape, rm)
    for i in range(n):
        assert np.array_equal(masks[i], g[f"mask{i}"].numpy())
    mine = prune.mask_generator_state_dict(full, masks)
    assert list(mine) == list(pruned)
    for k in pruned:
        assert mine[k].shape == pruned[k].shape and torch.equal(mine[k], pruned[k]), k
    c = load_json("contract_256")
    assert [int(x * 0.7) for x in c["full_shape"]] == [a - b for a, b in zip(c["full_shape"], c["pruned_shape"])]


def test_mac_counts_match_reference_constants():
    """Util/Calculators.py:13-14 — the reference's only known-answer constants."""
    c = load_json("contract_256")

    def macs(shape, keys):
        # the repo counts up-convs at their INPUT resolution: layer i runs at 2^(2 + i//2) (Calculators.py:5-9,29)
        conv = sum(shape[i] * shape[i + 1] * 9 * (2 ** (2 + i // 2)) ** 2 for i in range(len(shape) - 1))
        rgb = sum(shape[2 * i + 1] * 3 * (2 ** (2 + i)) ** 2 for i in range(len(shape) // 2))
        d = dict((k, s) for k, s in keys)
        mapping = sum(int(np.prod(s)) for k, s in d.items() if k.startswith("style") and k.endswith("weight"))
        mod = sum(int(np.prod(s)) for k, s in d.items() if k.endswith("modulation.weight"))
        return conv + rgb + mapping + mod
    assert macs(c["full_shape"], c["full_keys"]) == c["kat_macs_256"] == c["macs_full"]
    assert macs(c["pruned_shape"], c["pruned_keys"]) == c["macs_pruned"]


def _ddp_inputs(g, meta):
    """8 seeded samples; rank r of 2 owns samples r::2 so that the discriminator's minibatch-stddev groups
    (model.py:784-790: view(group=4, batch/4, ...) puts samples {i, i+2, i+4, i+6} of a batch of 8 together) are the
    same sets of samples in the 1-process and the 2-process run — the DataParallel->DDP drift of SURVEY.md §7 then
    vanishes and the two runs must agree to fp32 reduction-order accuracy."""
    gen = torch.Generator().manual_seed(77)
    nl = 7
    return dict(z=[torch.randn(8, 24, generator=gen) for _ in range(2)],
                mask=(torch.rand(8, 1, 32, 32, generator=gen) > 0.4).float(),
                sn=[torch.randn(8, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(nl)],
                tn=[torch.randn(8, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(nl)])


@pytest.mark.parametrize("tag", ["Intermediate", "Intermediate_percept", "Output_Only_percept"])
def test_kd_modes_match_reference(tag):
    """kd_mode='Intermediate' (train.py:165-169) and the LPIPS term's data flow (train.py:173-182, stand-in distance) vs
    the reference's own G_Loss_BackProp / KD_loss, with the content mask derived by cagc.content_mask from the same
    stand-in parsing logits the reference's Batch_Img_Parsing saw."""
    import torch.nn.functional as F
    g = load_npz("kd_modes_tiny")
    base = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _kd_objects(base, meta)
    B = 4
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    cls = torch.zeros(B, 512, 512, dtype=torch.long)
    for i in range(B):
        cls[i][((yy - 256 - 10 * i) / 200.0) ** 2 + ((xx - 256 + 7 * i) / 150.0) ** 2 < 1.0] = 1 + i
        cls[i][(yy > 440)] = 16
    logits = F.one_hot(cls, 19).permute(0, 3, 1, 2).float()
    mode = "Intermediate" if tag.startswith("Intermediate") else "Output_Only"
    percept = (lambda a, b: ((a - b) ** 2).mean(dim=[1, 2, 3])) if tag.endswith("percept") else None
    step = kd.KDStep(student, teacher, disc, latent=24, parsing_net=lambda x: (logits,), kd_mode=mode, percept_loss=percept)
    p = tag + "/"
    nl = student.num_layers
    n_z = int(g[p + "n_z"])
    inj = int(g[p + "inject_index"])
    losses = step.g_step([g[p + f"z{i}"] for i in range(n_z)], None if inj < 0 else inj, None,
                         student_noise=[g[p + f"student_noise{i}"] for i in range(nl)],
                         teacher_noise=[g[p + f"teacher_noise{i}"] for i in range(nl)])
    assert abs(losses["g"].item() - float(g[p + "g"])) < 3e-5
    assert abs(losses["kd_l1_loss"].item() - float(g[p + "kd_l1_loss"])) < 3e-5 * max(1.0, abs(float(g[p + "kd_l1_loss"])))
    if percept is not None:
        assert abs(losses["kd_lpips_loss"].item() - float(g[p + "kd_lpips_loss"])) < 3e-5
    params = dict(student.named_parameters())
    for k, v in sub(g, p + "grad/").items():
        assert_close(params[k].grad, v, 3e-4 if v.numel() > 1 else 3e-3, f"{tag} grad {k}")


def _ddp_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "content-aware-gan-compression_amd"), os.path.dirname(os.path.abspath(__file__))):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cagc import distributed as cd
    torch.set_num_threads(2)
    cd.init_from_env(backend="gloo")
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _kd_objects(g, meta)
    ddp = cd.wrap_student(student, torch.device("cpu"))
    step = kd.KDStep(ddp, teacher, disc, latent=24)
    d = _ddp_inputs(g, meta)
    sl = slice(rank, None, world)
    losses = step.g_step([z[sl] for z in d["z"]], 3, d["mask"][sl], student_noise=[n[sl] for n in d["sn"]],
                         teacher_noise=[n[sl] for n in d["tn"]])
    red = cd.reduce_loss_dict(losses)
    if rank == 0:
        torch.save({"params": {k: v.detach().clone() for k, v in student.named_parameters()},
                    "grads": {k: v.grad.clone() for k, v in student.named_parameters()},
                    "kd": red["kd_l1_loss"].item(), "g": red["g"].item()}, tmp)
    cd.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_equal_one_rank_on_concatenated_batch(tmp_path):
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    tmp = str(tmp_path / "ddp.pt")
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_ddp_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    student, teacher, disc = _kd_objects(g, meta)
    step = kd.KDStep(student, teacher, disc, latent=24)
    d = _ddp_inputs(g, meta)
    losses = step.g_step(d["z"], 3, d["mask"], student_noise=d["sn"], teacher_noise=d["tn"])
    assert abs(got["kd"] - losses["kd_l1_loss"].item()) < 1e-5 and abs(got["g"] - losses["g"].item()) < 1e-5
    for k, p in student.named_parameters():
        assert_close(got["grads"][k], p.grad, 2e-4 if p.numel() > 1 else 3e-3, "ddp grad " + k)
        assert_close(got["params"][k], p.detach(), 1e-4, "ddp param " + k)


def test_kd_step_without_content_mask_is_plain_l1():
    """Content-aware KD off (reference train.py:155, 516-518: parsing_net None): KD_loss is an un-masked L1 between the
    teacher's and the student's image.  KDStep(parsing_net=None) with mask=None must compute exactly that (advisor r2)."""
    g = load_npz("kd_step_tiny")
    meta = load_json("kd_step_tiny_meta")
    student, teacher, disc = _kd_objects(g, meta)
    step = kd.KDStep(student, teacher, disc, latent=24)
    st = meta["steps"][0]
    nl = student.num_layers
    zs = [g[f"step0/z{i}"] for i in range(st["n_z"])]
    sn = [g[f"step0/student_noise{i}"] for i in range(nl)]
    tn = [g[f"step0/teacher_noise{i}"] for i in range(nl)]
    g_loss, kd_l1, fake = step.g_losses(zs, st["inject_index"], None, sn, tn)
    with torch.no_grad():
        t_img = teacher(zs, inject_index=st["inject_index"], noise=tn)
    assert abs(kd_l1.item() - 3 * torch.mean(torch.abs(t_img - fake)).item()) < 1e-6
    ones = torch.ones(fake.shape[0], 1, fake.shape[2], fake.shape[3])
    _, kd_ones, _ = step.g_losses(zs, st["inject_index"], ones, sn, tn)
    assert abs(kd_l1.item() - kd_ones.item()) < 1e-6          # an all-ones mask is the same loss
    losses = step.g_step(zs, st["inject_index"], None, sn, tn)  # and the whole step runs (backward + Adam)
    assert torch.isfinite(losses["kd_l1_loss"]) and all(p.grad is not None for p in student.parameters() if p.requires_grad)
