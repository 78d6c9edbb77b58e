"""The student's ToRGB chain on a side stream (cagc/op/modconv.py _ToRGB, FORK_TORGB): the forked and the one-stream order must give
the same image and the same gradients — bit for bit in deterministic mode — for private chains (image only: the skip gradients stay on
the side stream) and for escaping RGB lists ('Intermediate' distillation reads every resolution), repeatedly (a race shows up as a
sporadic mismatch), eagerly and under HIP-graph capture.  Reference: model.py:380-395 (ToRGB), :545-666."""
import pytest
import torch

import cagc.model as M
from cagc import _lib, kd
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(net, z, noise, proj, want_list):
    net.zero_grad(set_to_none=True)
    out = net([z], noise=noise, return_rgb_list=want_list)
    imgs = out if want_list else [out]
    loss = sum((im * p).sum() for im, p in zip(imgs, proj[-len(imgs):]))
    loss.backward()
    torch.cuda.synchronize()
    return [im.detach().clone() for im in imgs], {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("want_list", [False, True], ids=["private_chain", "rgb_list_escapes"])
def test_forked_torgb_chain_equals_one_stream_order(monkeypatch, want_list):
    torch.manual_seed(5)
    B = 8
    net = M.Generator(256, 512, 2, generator_net_shape=[154] * 10 + [77, 77, 39, 39]).to(DEV)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    z = torch.randn(B, 512, device=DEV)
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV) for i in range(net.num_layers)]
    proj = [torch.randn(B, 3, 2 ** r, 2 ** r, device=DEV) for r in range(2, 9)]
    with _lib.tuning(deterministic=1):
        monkeypatch.setattr(mc, "FORK_TORGB", False)
        img0, g0 = _run(net, z, noise, proj, want_list)
        monkeypatch.setattr(mc, "FORK_TORGB", True)
        monkeypatch.setattr(mc, "FORK_TORGB_MODE", 2)
        monkeypatch.setattr(mc, "FORK_TORGB_MIN_BATCH", 1)
        for rep in range(6):
            junk = [torch.randn(1 << 22, device=DEV) for _ in range(3)]      # churn the caller's allocator pool between passes
            del junk
            img1, g1 = _run(net, z, noise, proj, want_list)
            for a, b in zip(img0, img1):
                assert torch.equal(a, b), f"image differs with the forked ToRGB chain (pass {rep})"
            assert g0.keys() == g1.keys()
            for k in g0:
                assert torch.equal(g0[k], g1[k]), f"gradient {k} differs with the forked ToRGB chain (pass {rep})"
    assert mc._fork_streams, "the fork never happened"


def test_forked_chain_inside_the_captured_step_equals_eager(monkeypatch):
    """GraphedKDStep with the fork on == the eager one-stream step on the same inputs (deterministic mode: bit-equal losses, gradients to
    rounding of the Adam update)."""
    torch.manual_seed(6)
    B = 8
    student, teacher, disc = kd.build_synthetic_workload(256, DEV, seed=3)
    mask = kd.ellipse_mask(B, 256, DEV)
    nl = student.num_layers
    zs = [torch.randn(B, 512, device=DEV), torch.randn(B, 512, device=DEV)]
    sn = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV) for i in range(nl)]
    tn = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV) for i in range(nl)]
    import copy
    with _lib.tuning(deterministic=1):
        monkeypatch.setattr(mc, "FORK_TORGB", False)
        s0 = copy.deepcopy(student)
        e = kd.KDStep(s0, teacher, disc)
        l0 = e.g_step(zs, 3, mask, sn, tn)
        monkeypatch.setattr(mc, "FORK_TORGB", True)
        monkeypatch.setattr(mc, "FORK_TORGB_MODE", 2)
        monkeypatch.setattr(mc, "FORK_TORGB_MIN_BATCH", 1)
        s1 = copy.deepcopy(student)
        g = kd.GraphedKDStep(s1, teacher, disc, B, mask, random_noise=False)
        l1 = g.g_step(zs, 3, mask, sn, tn)
        torch.cuda.synchronize()
    assert float(l0["g"]) == float(l1["g"]) and float(l0["kd_l1_loss"]) == float(l1["kd_l1_loss"])
    p0, p1 = dict(s0.named_parameters()), dict(s1.named_parameters())
    for k in p0:      # deterministic mode: the gradients of the forked captured step equal the eager one-stream step's bit for bit
        g0, g1 = p0[k].grad, p1[k].grad
        if k.startswith("style."):     # (the mapping network's weight gradients: to the last bits — tests/test_gpu_parity.py _graph_vs_eager_steps)
            assert float((g0 - g1).abs().max()) <= 2e-6 * float(g0.abs().max()), k
        else:
            assert torch.equal(g0, g1), f"gradient {k} differs by {float((g0 - g1).abs().max()):.3e}"
    worst = max(float((p0[k].detach() - p1[k].detach()).abs().max() / p0[k].detach().abs().max().clamp(min=1e-30)) for k in p0)
    assert worst <= 1e-4, f"updated weights differ between the eager one-stream step and the forked captured step: {worst:.2e}"



def test_graphed_step_builds_on_a_student_left_frozen_by_a_d_step():
    """bench.py's legs found this: TrainIteration.d_step leaves the student with requires_grad False; a GraphedKDStep built afterwards
    (on the same student or a deepcopy) must un-freeze it before it registers its gradient hooks."""
    student, teacher, disc = kd.build_synthetic_workload(256, DEV, seed=5)
    mask = kd.ellipse_mask(2, 256, DEV)
    kd.requires_grad(student, False)
    step = kd.GraphedKDStep(student, teacher, disc, 2, mask)
    out = step.sample_and_step()
    torch.cuda.synchronize()
    assert torch.isfinite(out["g"]) and all(p.requires_grad for p in student.parameters())
