"""the content-aware saliency score of prune.py, vectorised on device, against a golden produced by
the reference's own Get_Salt_Pepper_Noisy_Image + Get_Weight_Gradient (Util/content_aware_pruning.py:152-196)."""
import pytest
import torch

import cagc.model as M
from cagc import prune
from oracle import ref_model
from _util import assert_close, load_json, load_npz, sub


def _run(dev, tol):
    g = load_npz("saliency_tiny")
    meta = load_json("generator_tiny_keys")
    net = M.Generator(meta["config"]["size"], meta["config"]["style_dim"], meta["config"]["n_mlp"],
                      generator_net_shape=meta["config"]["shape"])
    net.load_state_dict(sub(g, "sd/"), strict=True)
    net = net.to(dev)
    noise = [g[f"noise{i}"].to(dev) for i in range(net.num_layers)]
    img = net([g["z"].to(dev)], noise=noise)
    scores = prune.batch_saliency_scores(net, img, g["hit"].to(dev), g["sp"].to(dev))
    assert len(scores) == g["n_layers"] == len(prune.network_shape(sub(g, "sd/")))
    for i, sc in enumerate(scores):
        assert_close(sc, g[f"score{i}"], tol, f"layer {i} score")


def test_saliency_scores_cpu_match_reference():
    _run("cpu", 5e-5)


@pytest.mark.gpu
def test_saliency_scores_gpu_match_reference():
    _run("cuda", 1e-3)


def test_oracle_saliency_matches_reference():
    g = load_npz("saliency_tiny")
    sd = sub(g, "sd/")
    names = ["conv1.conv.weight"] + [k for k in sd if k.startswith("convs.") and k.endswith(".conv.weight")]
    names.append(sorted(k for k in sd if k.startswith("to_rgbs.") and k.endswith("conv.weight"))[-1])
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sdr = dict(sd)
    sdr.update(leaves)
    img = ref_model.generator_forward_ref(sdr, [g["z"]], noise=[g[f"noise{i}"] for i in range(7)])
    loss = (g["hit"] * (g["sp"] - img).abs()).sum()
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    for i, gr in enumerate(grads):
        assert_close(gr.abs().mean(dim=[0, 1, 3, 4]), g[f"score{i}"], 5e-5, f"oracle layer {i}")


def test_salt_pepper_statistics():
    torch.manual_seed(0)
    mask = torch.zeros(8, 1, 64, 64)
    mask[:, :, 16:48, 16:48] = 1
    hit, sp = prune.salt_pepper(mask, 0.25)
    assert float((hit * (1 - mask)).sum()) == 0                       # never outside the mask
    frac = float(hit.sum() / mask.sum())
    assert 0.22 < frac < 0.28
    assert set(sp[hit > 0].unique().tolist()) == {-1.0, 1.0} and float(sp[hit == 0].abs().sum()) == 0


# ---------------------------------------------------------------------------------------------------
# sharded sweep: batches dealt round-robin to 2 gloo ranks + one all-reduce == the single-process sweep
# ---------------------------------------------------------------------------------------------------
def _sweep_objects():
    g = load_npz("saliency_tiny")
    meta = load_json("generator_tiny_keys")
    net = M.Generator(meta["config"]["size"], meta["config"]["style_dim"], meta["config"]["n_mlp"],
                      generator_net_shape=meta["config"]["shape"])
    net.load_state_dict(sub(g, "sd/"), strict=True)
    size = meta["config"]["size"]

    def mask_fn(img):
        m = torch.zeros(img.shape[0], 1, size, size)
        m[:, :, 6:26, 8:24] = 1
        return m

    def noise_fn(idx, b):
        gen = torch.Generator().manual_seed(1000 + idx)
        return [torch.randn(b, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(net.num_layers)]

    return net, mask_fn, noise_fn, meta["config"]["style_dim"]


def _sweep_worker(rank, world, port, tmp):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cagc import distributed as cd
    torch.set_num_threads(2)
    cd.init_from_env(backend="gloo")
    net, mask_fn, noise_fn, sdim = _sweep_objects()
    sc = prune.content_aware_scores(net, 5 * 3, 3, 0.3, mask_fn, torch.device("cpu"), latent_dim=sdim, rank=rank, world=world,
                                    seed=77, noise_fn=noise_fn)
    if rank == 0:
        torch.save([t.clone() for t in sc], tmp)
    cd.barrier()
    dist.destroy_process_group()


def test_sharded_saliency_sweep_two_ranks_equal_one_rank(tmp_path):
    import os
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "sweep.pt")
    mp.spawn(_sweep_worker, args=(2, 29700 + (os.getpid() % 200), tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    net, mask_fn, noise_fn, sdim = _sweep_objects()
    want = prune.content_aware_scores(net, 5 * 3, 3, 0.3, mask_fn, torch.device("cpu"), latent_dim=sdim, seed=77, noise_fn=noise_fn)
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert_close(a, b, 1e-6, f"sharded score layer {i}")
