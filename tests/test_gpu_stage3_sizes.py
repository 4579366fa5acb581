"""Parity at the sizes the reference's stage 3 really composites at (outpainting_updown_sampling_redux.py:72-82, 104-105, 403-458: sides in
[1024, 2800]; UODD is up-scaled to 2048): joint sequences of 17 625 tokens (2048 x 2048) and 24 166 tokens (2800 x 2096) — where attention is
more than half of a block's FLOPs — against the CPU oracle, with the bars of the 1024^2 tests (tests/test_gpu_fullsize.py)."""
import math
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("h,w", [(128, 128), (131, 175)])
def test_stage3_size_flux_blocks_vs_oracle(gpu, h, w):
    """1 double + 1 single block of the FLUX.1-Fill-dev width (D = 3072, 24 heads) on the token grid of a 2048 x 2048 and of a
    2800 x 2096 composite (B = 1, 1241 text + Redux tokens): HIP vs the bf16 oracle and the float32 oracle on identical weights"""
    from domain_rag_amd.flux import FluxTransformerHIP, latent_image_ids
    from domain_rag_amd.flux_params import FluxConfig, init_params
    from oracle import flux as oflux
    cfg = FluxConfig(in_channels=384, num_layers=1, num_single_layers=1)
    params = init_params(cfg, seed=21)
    g = torch.Generator().manual_seed(22 + h)
    St = 512 + 729
    hidden = torch.randn(1, h * w, 384, generator=g).bfloat16()
    enc = torch.randn(1, St, 4096, generator=g).bfloat16()
    pooled = torch.randn(1, 768, generator=g).bfloat16()
    t, gd = torch.tensor([0.6172]), torch.tensor([30.0])
    img_ids, txt_ids = latent_image_ids(h, w), torch.zeros(St, 3)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    taps_ref, taps32, taps = {}, {}, {}
    p32 = {k: v.float() for k, v in params.items()}
    t0 = time.perf_counter()
    from conftest import oracle_threads
    with torch.no_grad():
        oracle_threads(torch.bfloat16)
        ref = oflux.flux_forward(params, ocfg, hidden, enc, pooled, t, img_ids, txt_ids, gd, taps=taps_ref)
        oracle_threads(torch.float32)
        ref32 = oflux.flux_forward(p32, ocfg, hidden.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, gd, taps=taps32,
                                   time_dtype=torch.bfloat16)
    t_or = time.perf_counter() - t0
    out = FluxTransformerHIP(cfg, params, gpu)(hidden.to(gpu), enc.to(gpu), pooled.to(gpu), t, img_ids, txt_ids, gd, taps=taps)
    print(f"\n[stage-3 size {h}x{w} tokens, S = {St + h * w}] oracle (bf16 + f32) {t_or:.1f} s on the host")
    for name, got, rbf, r32 in (("double.0", taps["double.0"], taps_ref["double.0"], taps32["double.0"]),
                                ("single.0", taps["single.0"], taps_ref["single.0"], taps32["single.0"]), ("out", out, ref, ref32)):
        e, e_or = _rel(got, r32), _rel(rbf, r32)
        print(f"    {name}: HIP vs f32 {e:.3e}   bf16 oracle vs f32 {e_or:.3e}   ratio {e / max(e_or, 1e-30):.2f}")
        assert e < max(1e-2, 1.3 * e_or), f"{name}: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"
        assert _rel(got, rbf) < 2e-2, name


def test_attention_alone_at_the_resolution_cap(gpu):
    """S = 24 166 (2800 x 2096), 24 heads of 128: the attention kernel alone against torch's scaled_dot_product_attention in float32 on
    the same bf16 q, k, v (4 heads compared: the CPU side is 1.2 TFLOP per head pair), and against the same call in bf16 for the bar"""
    from domain_rag_amd import ops
    B, S, H = 1, 131 * 175 + 1241, 24
    D = H * 128
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, S, 3 * D, generator=g).bfloat16()
    d = qkv.to(gpu)
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, dtype=torch.bfloat16, device=gpu)
    ops.qk_norm_rope_vt(d, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
    o = torch.empty(B, S, D, dtype=torch.bfloat16, device=gpu)
    ops.attention(d, d.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
    o = o.cpu().view(B, S, H, 128)
    assert torch.isfinite(o.float()).all()
    heads = [0, 7, 16, 23]
    q, k, v = (qkv[..., i * D:(i + 1) * D].view(B, S, H, 128)[:, :, heads].transpose(1, 2) for i in range(3))
    with torch.no_grad():
        r32 = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
        rbf = F.scaled_dot_product_attention(q, k, v)
    got = o[:, :, heads].transpose(1, 2)
    e, e_or = _rel(got, r32), _rel(rbf, r32)
    print(f"\n[attention S = {S}] HIP vs f32 {e:.3e}   torch bf16 vs f32 {e_or:.3e}")
    assert e < max(1e-2, 1.3 * e_or), (e, e_or)
