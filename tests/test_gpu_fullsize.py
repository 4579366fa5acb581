"""Parity at BASELINE's FULL sizes (S = 1241 + 4096 = 5337 tokens, D = 3072, 24 heads; corpus N = 118 287):
one real-size Flux block against the CPU oracle, and size-independent properties where the oracle is too slow."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_full_size_flux_blocks_vs_oracle(gpu):
    """1 double + 1 single block of FLUX.1-Fill-dev at 1024^2 (B=1): HIP vs the bf16 CPU oracle on identical weights"""
    from domain_rag_amd.flux import FluxTransformerHIP, latent_image_ids
    from domain_rag_amd.flux_params import FluxConfig, init_params
    from oracle import flux as oflux
    cfg = FluxConfig(in_channels=384, num_layers=1, num_single_layers=1)
    params = init_params(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    St, h, w = 512 + 729, 64, 64
    hidden = torch.randn(1, h * w, 384, generator=g).bfloat16()
    enc = torch.randn(1, St, 4096, generator=g).bfloat16()
    pooled = torch.randn(1, 768, generator=g).bfloat16()
    t, gd = torch.tensor([0.6172]), torch.tensor([30.0])
    img_ids, txt_ids = latent_image_ids(h, w), torch.zeros(St, 3)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    taps_ref, taps32, taps = {}, {}, {}
    p32 = {k: v.float() for k, v in params.items()}
    from conftest import oracle_threads
    with torch.no_grad():
        oracle_threads(torch.bfloat16)
        ref = oflux.flux_forward(params, ocfg, hidden, enc, pooled, t, img_ids, txt_ids, gd, taps=taps_ref)
        oracle_threads(torch.float32)
        ref32 = oflux.flux_forward(p32, ocfg, hidden.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, gd, taps=taps32,
                                   time_dtype=torch.bfloat16)
    out = FluxTransformerHIP(cfg, params, gpu)(hidden.to(gpu), enc.to(gpu), pooled.to(gpu), t, img_ids, txt_ids, gd, taps=taps)
    # bar (round 3): the HIP path sits at most 1.3 x as far from float32 as the reference-dtype (bf16) oracle does (floor 1e-2, north_star's
    # figure); the fixed 2e-2 against the bf16 oracle stays as a second, absolute check
    for name, got, rbf, r32 in (("double.0", taps["double.0"], taps_ref["double.0"], taps32["double.0"]),
                                ("single.0", taps["single.0"], taps_ref["single.0"], taps32["single.0"]), ("out", out, ref, ref32)):
        e, e_or = _rel(got, r32), _rel(rbf, r32)
        assert e < max(1e-2, 1.3 * e_or), f"{name}: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"
        assert _rel(got, rbf) < 2e-2, name


def test_full_size_attention_properties(gpu):
    """S = 5337, 24 heads: (1) rows are convex combinations -> constant V gives exactly that constant;
    (2) permuting the keys (and values alike) leaves the output unchanged up to rounding; (3) a one-hot-dominant
    key row selects its value row."""
    from domain_rag_amd import ops
    B, S, H = 1, 5337, 24
    D = H * 128
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, S, 3 * D, generator=g).bfloat16()
    s_pad = (S + 63) // 64 * 64
    scale = 1 / math.sqrt(128)

    def run(t):
        d = t.to(gpu).clone()
        vt = torch.empty(B, H, 128, s_pad, dtype=torch.bfloat16, device=gpu)
        ops.qk_norm_rope_vt(d, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
        o = torch.empty(B, S, D, dtype=torch.bfloat16, device=gpu)
        ops.attention(d, d.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, scale)
        return o.cpu()

    c = qkv.clone(); c[..., 2 * D:] = 0.75
    assert torch.equal(run(c), torch.full((B, S, D), 0.75, dtype=torch.bfloat16))
    base = run(qkv)
    perm = torch.randperm(S, generator=g)
    pq = qkv.clone(); pq[:, :, D:] = qkv[:, perm, D:]            # permute k and v rows together
    assert _rel(run(pq), base) < 2e-2
    sp = qkv.clone()
    sp[0, 4321, D:2 * D] = sp[0, 17, 0:D] * 4.0                  # key 4321 aligned with query 17 in every head
    o = run(sp)
    assert _rel(o[0, 17], sp[0, 4321, 2 * D:]) < 3e-2


def test_corpus_scale_topk_properties(gpu):
    """N = 118 287 x 512 (242 MB): planted neighbours come back first, scores descend, results equal the oracle on a
    sampled query, the call is deterministic.  (Sharded all-gather order == single order at this size:
    tests/test_gpu_fullsize2.py::test_sharded_gather_order_equals_single_order.)"""
    from domain_rag_amd import ops
    from oracle import retrieval as oret
    N = 118287
    g = torch.Generator(device=gpu).manual_seed(1)
    corpus = torch.randn(N, 512, generator=g, device=gpu)
    corpus /= corpus.norm(dim=-1, keepdim=True)                  # test-data preparation (torch), not the product path
    plant = torch.tensor([5, 60000, 118286, 777, 99999], device=gpu)
    q = corpus[plant] + 0.02 * torch.randn(5, 512, generator=g, device=gpu)
    D, I = ops.cosine_topk(corpus, q, 100)
    Dc, Ic = D.cpu().numpy(), I.cpu().numpy()
    assert (Ic[:, 0] == plant.cpu().numpy()).all()
    assert (Dc[:, :-1] >= Dc[:, 1:]).all() and (Ic >= 0).all() and (Ic < N).all()
    assert all(len(set(r)) == 100 for r in Ic)
    Dr, Ir = oret.cosine_topk(corpus.cpu().numpy(), q[:2].cpu().numpy(), 100)
    assert np.array_equal(Ic[:2], Ir) and np.array_equal(Dc[:2], Dr)
    D2, I2 = ops.cosine_topk(corpus, q, 100)                     # idempotent / deterministic
    assert torch.equal(I2, I) and torch.equal(D2, D)


def test_topk_past_the_candidate_workspace_bound(gpu):
    """N = 2.1 M rows (4.3 GB): 64 candidate regions with room for every row would pass 1 GiB, so a pass carries 32 queries and
    Q = 40 takes two passes; results equal one-query calls (which take the plain path) and, on two queries, the oracle"""
    from domain_rag_amd import ops
    from oracle import retrieval as oret
    N, Q = 2_100_000, 40
    g = torch.Generator(device=gpu).manual_seed(5)
    corpus = torch.randn(N, 512, generator=g, device=gpu)
    q = torch.randn(Q, 512, generator=g, device=gpu)
    D, I = ops.cosine_topk(corpus, q, 100)
    for j in (0, 31, 32, 39):
        d1, i1 = ops.cosine_topk(corpus, q[j:j + 1].contiguous(), 100)
        assert torch.equal(D[j:j + 1], d1) and torch.equal(I[j:j + 1], i1), j
    Dr, Ir = oret.cosine_topk(corpus.cpu().numpy(), q[[3, 37]].cpu().numpy(), 100)
    assert np.array_equal(I[[3, 37]].cpu().numpy(), Ir) and np.array_equal(D[[3, 37]].cpu().numpy(), Dr)
