"""on-device content mask.  Golden = the reference's own Batch_Img_Parsing + Get_Masked_Tensor
(Util/content_aware_pruning.py:61-117) around a stand-in parsing net (oracle/gen_golden.py::gold_content_mask).
The mask is integer / dyadic work: BIT-exact for the oracle, the product's CPU path and the HIP kernels."""
import numpy as np
import pytest
import torch

from _util import load_npz
from cagc import content_mask as cm
from oracle import ref_content_mask as rcm
from oracle import synth

SIZES = (32, 256, 1024)
B = 2


@pytest.fixture(scope="module")
def gold():
    return load_npz("content_mask")


@pytest.fixture(scope="module")
def logits(gold):
    lg = synth.parsing_logits(B, 512, 19, seed=1)
    chk = np.array([lg.astype(np.float64).sum(), np.abs(lg.astype(np.float64)).sum()])
    assert np.allclose(chk, gold["logits_checksum"].numpy(), rtol=1e-12), "synthetic logits differ from the fixture's"
    return lg


def _img(S):
    return synth.hash01((B, 3, S, S), seed=100 + S) * np.float32(2.6) - np.float32(1.3)


def _gold_mask(gold, S):
    bits = np.unpackbits(gold[f"S{S}/mask_bits"].numpy())[: B * S * S]
    return bits.reshape(B, 1, S, S).astype(np.float32)


def test_oracle_content_mask_bit_exact(gold, logits):
    parsing = rcm.parsing_ref(logits)
    assert int(parsing.sum()) == int(gold["parsing_sum"][0])
    assert int((parsing * np.arange(512)[None, None, :]).sum()) == int(gold["parsing_sum"][1])
    for S in SIZES:
        m = rcm.content_mask_ref(parsing, S)
        assert np.array_equal(m, _gold_mask(gold, S)), f"oracle mask differs at S={S}"
        assert int(m.sum()) == int(gold[f"S{S}/mask_sum"])


def test_oracle_parsing_input(gold):
    for S in SIZES:
        x = rcm.parsing_input_ref(_img(S))
        assert np.abs(x[:, :, ::7, ::5] - gold[f"S{S}/parse_in_sub"].numpy()).max() <= 2e-6
        assert abs(x.astype(np.float64).sum() - float(gold[f"S{S}/parse_in_sum"][0])) <= 1e-6 * float(gold[f"S{S}/parse_in_sum"][1])


def test_product_cpu_path(gold, logits):
    lg = torch.from_numpy(logits)
    for S in SIZES:
        assert np.array_equal(cm.content_mask(lg, S).numpy(), _gold_mask(gold, S))
        x = cm.parsing_input(torch.from_numpy(_img(S)))
        assert (x[:, :, ::7, ::5] - gold[f"S{S}/parse_in_sub"]).abs().max().item() <= 2e-6
    # batch 1 (the reference's .squeeze() breaks there): batch independence
    assert np.array_equal(cm.content_mask(lg[1:2], 256).numpy(), _gold_mask(gold, 256)[1:2])


def test_kd_step_derives_mask_from_parsing_net_cpu():
    """KDStep(parsing_net=...) == KDStep with the same mask supplied explicitly (tiny models, CPU)."""
    import cagc.model as M
    from cagc import kd
    torch.manual_seed(3)
    student = M.Generator(32, 24, 2, generator_net_shape=[5, 5, 4, 4, 3, 3, 2, 2])
    teacher = M.Generator(32, 24, 2, generator_net_shape=[11, 11, 7, 7, 5, 5, 3, 3])
    disc = M.Discriminator(32)
    lg = torch.from_numpy(synth.parsing_logits(2, 512, 19, seed=5))
    zs = [torch.randn(2, 24)]
    nl = student.num_layers
    sn = [torch.randn(2, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)) for i in range(nl)]
    tn = [torch.randn(2, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)) for i in range(nl)]
    calls = []

    def net(x):
        calls.append(tuple(x.shape))
        return (lg,)

    a = kd.KDStep(student, teacher, disc, latent=24, parsing_net=net)
    la = a.g_losses(zs, None, None, sn, tn)
    assert calls == [(2, 3, 512, 512)]
    b = kd.KDStep(student, teacher, disc, latent=24)
    lb = b.g_losses(zs, None, cm.content_mask(lg, 32), sn, tn)
    assert torch.equal(la[1], lb[1]) and torch.equal(la[0], lb[0])


@pytest.mark.gpu
def test_hip_content_mask_bit_exact(gold, logits):
    dev = torch.device("cuda")
    lg = torch.from_numpy(logits).to(dev)
    for S in SIZES:
        m = cm.content_mask(lg, S)
        assert m.is_cuda and tuple(m.shape) == (B, 1, S, S)
        assert np.array_equal(m.cpu().numpy(), _gold_mask(gold, S)), f"HIP mask differs at S={S}"
        x = cm.parsing_input(torch.from_numpy(_img(S)).to(dev))
        assert (x[:, :, ::7, ::5].cpu() - gold[f"S{S}/parse_in_sub"]).abs().max().item() <= 2e-6
        assert abs(x.double().sum().item() - float(gold[f"S{S}/parse_in_sum"][0])) <= 1e-6 * float(gold[f"S{S}/parse_in_sum"][1])
    assert np.array_equal(cm.content_mask(lg[1:2], 256).cpu().numpy(), _gold_mask(gold, 256)[1:2])
    # unaligned logits pointer (scalar kernel) and ties (first maximum wins, as torch.argmax)
    flat = torch.empty(lg.numel() + 1, device=dev)
    flat[1:] = lg.reshape(-1)
    assert np.array_equal(cm.content_mask(flat[1:].view_as(lg), 256).cpu().numpy(), _gold_mask(gold, 256))
    ties = torch.zeros(1, 19, 512, 512, device=dev)
    ties[:, 3] = 1.0
    ties[:, 7] = 1.0
    ties[:, 16, 300:] = 1.0           # tie between 3, 7 and 16: class 3 wins -> kept everywhere
    assert cm.content_mask(ties, 256).min().item() == 1.0
    ties[:, 16, 300:] = 2.0
    assert torch.equal(cm.content_mask(ties, 256).cpu(), cm.content_mask(ties.cpu(), 256))


@pytest.mark.gpu
def test_hip_content_mask_full_batch_matches_oracle():
    """bs 16 (the KD step's batch) vs the numpy oracle, 256 px."""
    dev = torch.device("cuda")
    lg = synth.parsing_logits(16, 512, 19, seed=9)
    want = rcm.content_mask_ref(rcm.parsing_ref(lg), 256)
    got = cm.content_mask(torch.from_numpy(lg).to(dev), 256).cpu().numpy()
    assert np.array_equal(got, want)
