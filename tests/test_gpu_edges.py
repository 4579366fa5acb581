"""Edge cases through the C ABI: minimal / ragged / maximal shapes, NaNs, degenerate sequences, argument errors."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _randn(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).bfloat16()


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (1, 3072, 3072), (7, 12, 128), (129, 132, 64), (2047, 252, 256), (2048, 256, 256),
                                   (2303, 260, 320), (65536 + 3, 64, 64)])
def test_gemm_extreme_shapes(gpu, M, N, K):
    from domain_rag_amd import ops
    a, w, b = _randn((M, K), 1), _randn((N, K), 2, 0.05), _randn((N,), 3)
    out = ops.gemm(a.to(gpu), w.to(gpu), bias=b.to(gpu)).cpu()
    assert _rel(out, a.double() @ w.double().T + b.double()) < 6e-3


def test_gemm_argument_errors(gpu):
    from domain_rag_amd import ops
    a, w = _randn((8, 64), 1).to(gpu), _randn((8, 64), 2).to(gpu)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.gemm(a, _randn((6, 64), 3).to(gpu))
    with pytest.raises(RuntimeError, match="gate needs resid"):
        ops.gemm(a, w, gate=_randn((1, 8), 4).to(gpu), ldg=8)
    with pytest.raises(TypeError):
        ops.gemm(a.float(), w)


@pytest.mark.parametrize("S", [1, 31, 63, 64, 65, 127, 128, 129, 257])
def test_attention_sequence_boundaries(gpu, S):
    from domain_rag_amd import ops
    from oracle import ops_ref
    B, H = 2, 2
    D = H * 128
    qkv = _randn((B, S, 3 * D), S)
    s_pad = (S + 63) // 64 * 64
    d = qkv.to(gpu).clone()
    vt = torch.full((B, H, 128, s_pad), float("nan"), dtype=torch.bfloat16, device=gpu)
    ops.qk_norm_rope_vt(d, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
    assert torch.equal(d.cpu(), qkv)                                   # no norm / rope requested: q, k untouched
    assert torch.isfinite(vt.float()).all()                            # pad keys written as zeros
    out = torch.full((B, S, D), float("nan"), dtype=torch.bfloat16, device=gpu)
    scale = 1 / math.sqrt(128)
    ops.attention(d, d.view(-1)[D:], vt, out, B, S, H, 3 * D, S * 3 * D, D, S * D, scale)
    q, k, v = [t.view(B, S, H, 128).transpose(1, 2) for t in qkv.split(D, dim=-1)]
    assert _rel(out, ops_ref.attention_ref_f64(q, k, v, scale)) < 1.5e-2


def test_attention_huge_logits_stay_finite(gpu):
    """|q.k| ~ 1e4 after scaling: the exp2-domain softmax with deferred rescaling must not overflow"""
    from domain_rag_amd import ops
    from oracle import ops_ref
    B, S, H = 1, 200, 1
    qkv = _randn((B, S, 384), 3)
    qkv[..., :256] *= 30.0
    d = qkv.to(gpu).clone()
    vt = torch.empty((B, H, 128, 256), dtype=torch.bfloat16, device=gpu)
    ops.qk_norm_rope_vt(d, vt, None, None, None, None, None, None, B, S, H, 384, 0)
    out = torch.empty((B, S, 128), dtype=torch.bfloat16, device=gpu)
    ops.attention(d, d.view(-1)[128:], vt, out, B, S, H, 384, S * 384, 128, S * 128, 1 / math.sqrt(128))
    q, k, v = [t.view(B, S, H, 128).transpose(1, 2) for t in qkv.split(128, dim=-1)]
    assert torch.isfinite(out.float()).all()
    assert _rel(out, ops_ref.attention_ref_f64(q, k, v, 1 / math.sqrt(128))) < 2e-2


@pytest.mark.parametrize("N,d,Q,k", [(1, 64, 1, 1), (5, 1024, 2, 5), (4096, 64, 1, 2048), (4097, 128, 3, 1), (100000, 64, 17, 10),
                                     (20000, 1024, 40, 50), (9000, 256, 64, 128), (8208, 192, 65, 3)])
def test_topk_extremes(gpu, N, d, Q, k):
    from domain_rag_amd import ops
    from oracle import retrieval as oret
    rng = np.random.default_rng(N + d)
    c = rng.standard_normal((N, d)).astype(np.float32)
    q = rng.standard_normal((Q, d)).astype(np.float32)
    D, I = ops.cosine_topk(torch.from_numpy(c).to(gpu), torch.from_numpy(q).to(gpu), k)
    Dr, Ir = oret.cosine_topk(c, q, k)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))


def test_topk_nan_inf_and_zero_rows(gpu):
    """NaN scores rank last, +-inf order correctly, all-zero corpus = all ties -> index order"""
    from domain_rag_amd import ops
    from oracle import retrieval as oret
    rng = np.random.default_rng(7)
    c = rng.standard_normal((300, 64)).astype(np.float32)
    c[10, 3] = np.nan; c[20, :] = np.inf; c[30, :] = -np.inf; c[40:50] = 0.0
    q = np.abs(rng.standard_normal((2, 64))).astype(np.float32) + 0.1
    D, I = ops.cosine_topk(torch.from_numpy(c).to(gpu), torch.from_numpy(q).to(gpu), 300)
    Dr, Ir = oret.cosine_topk(c, q, 300)
    assert np.array_equal(I.cpu().numpy(), Ir)
    assert I[0, 0].item() == 20 and I[0, -1].item() == 10 and I[0, -2].item() == 30        # +inf first, -inf, then NaN last
    z = np.zeros((70, 64), np.float32)
    D0, I0 = ops.cosine_topk(torch.from_numpy(z).to(gpu), torch.from_numpy(q).to(gpu), 70)
    assert (I0.cpu().numpy() == np.arange(70)).all() and (D0.cpu().numpy() == 0).all()
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.cosine_topk(torch.zeros(10, 48, device=gpu), torch.zeros(1, 48, device=gpu), 1)
    with pytest.raises(RuntimeError, match="2048"):
        ops.cosine_topk(torch.zeros(10, 64, device=gpu), torch.zeros(1, 64, device=gpu), 4096)


def test_layernorm_and_groupnorm_constant_rows(gpu):
    """zero-variance inputs: eps keeps the result finite and equal to the bias / shift"""
    from domain_rag_amd import ops
    x = torch.full((4, 3072), 3.0).bfloat16()
    y = torch.empty((4, 3072), dtype=torch.bfloat16, device=gpu)
    g, b = _randn((3072,), 1), _randn((3072,), 2)
    ops.layernorm(x.to(gpu), y, 4, 3072, gamma=g.to(gpu), beta=b.to(gpu))
    assert torch.equal(y.cpu(), b.expand(4, -1))
    xc = torch.full((1, 8, 8, 128), -2.0).bfloat16()
    yc = torch.zeros((1, 10, 10, 128), dtype=torch.bfloat16, device=gpu)
    ops.groupnorm_silu(xc.to(gpu), yc, g[:128].to(gpu), b[:128].to(gpu), 1, 8, 8, 128, out_pad=1, silu=False)
    assert torch.equal(yc.cpu()[0, 1:-1, 1:-1], b[:128].expand(8, 8, -1))
