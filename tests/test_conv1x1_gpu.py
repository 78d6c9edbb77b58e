"""-m gpu: the per-image 1x1 GEMM kernel (csrc/conv1x1.hip, `cagc_gemm1x1`) — the discriminator ResBlock's skip conv and its data
gradient (reference model.py:724-737) — through the C ABI against float64: both workgroup shapes, ragged K / M / pixel counts,
the (alpha, beta) residual epilogue, and the transposed packing (A = W^T) against the autograd gradient of the forward."""
import pytest
import torch

from cagc import _lib
from cagc.op import modconv as mc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("cfg", [(2, 128, 256, 128 * 128), (16, 256, 512, 64 * 64), (4, 512, 512, 32 * 32), (16, 512, 512, 256),
                                 (3, 20, 36, 64), (1, 22, 77, 100), (2, 7, 5, 4), (1, 130, 200, 1028)])
@pytest.mark.parametrize("residual", [False, True])
def test_gemm1x1_forward_and_data_gradient_vs_float64(cfg, residual):
    B, K, M, P = cfg
    torch.manual_seed(61)
    w = torch.randn(M, K)
    x = torch.randn(B, K, P)
    r = torch.randn(B, M, P) if residual else None
    alpha, beta, scale = 0.7071, 0.5, 0.37
    ref = alpha * torch.matmul((w.double() * scale), x.double()) + (beta * r.double() if residual else 0)
    ap = mc.pack_gemm1x1(w.to(DEV), scale, False)
    xg = x.to(DEV)
    out = torch.full((B, M, P), float("nan"), device=DEV)
    _lib.call("cagc_gemm1x1", _lib.ptr(out), _lib.ptr(xg), _lib.ptr(ap), _lib.ptr(r.to(DEV)) if residual else None, B, K, M, P, alpha, beta)
    assert _rel(out, ref) <= 5e-6, _rel(out, ref)
    # data gradient: gx[b] = (scale W)^T @ g[b] — the same kernel on the transposed packing of the SAME weight tensor
    g = torch.randn(B, M, P)
    apt = mc.pack_gemm1x1(w.to(DEV), scale, True)
    gx = torch.full((B, K, P), float("nan"), device=DEV)
    _lib.call("cagc_gemm1x1", _lib.ptr(gx), _lib.ptr(g.to(DEV)), _lib.ptr(apt), None, B, M, K, P, 1.0, 0.0)
    ref_gx = torch.matmul((w.double() * scale).t(), g.double())
    assert _rel(gx, ref_gx) <= 5e-6, _rel(gx, ref_gx)


def test_gemm1x1_argument_contract():
    lib = _lib.load()
    x = torch.zeros(1, 8, 6, device=DEV)
    ap = torch.zeros(_lib.query("cagc_gemm1x1_packed_elems", 8, 8), device=DEV)
    out = torch.zeros(1, 8, 6, device=DEV)
    rc = lib.cagc_gemm1x1(_lib.ptr(out), _lib.ptr(x), _lib.ptr(ap), None, 1, 8, 8, 6, 1.0, 0.0, None)     # P % 4 != 0
    assert rc != 0 and b"multiple of 4" in lib.cagc_last_error()
    assert lib.cagc_gemm1x1(None, _lib.ptr(x), _lib.ptr(ap), None, 1, 8, 8, 8, 1.0, 0.0, None) != 0
