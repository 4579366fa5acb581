"""SigLIP / CLIP ViT encoders and the Redux prior on the HIP path vs the transformers-based oracles."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_siglip_shape_encoder(gpu):
    from domain_rag_amd import vit
    from oracle import vit as ov
    # head_dim 96 (padded to 128), intermediate 304 (padded to 320), patch K 588 (padded to 640)
    cfg = vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=3, intermediate=304)
    g = vit.init_generic_params(cfg, 1)
    img = (torch.rand(3, 56, 56, 3, generator=torch.Generator().manual_seed(0)) * 255).to(torch.uint8)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    ref32 = ov.siglip_last_hidden_state(g, 56, 14, 192, 2, 3, 304, px, torch.float32)
    refbf = ov.siglip_last_hidden_state(g, 56, 14, 192, 2, 3, 304, px, torch.bfloat16)
    out = vit.VitHIP(cfg, g, gpu)(img.to(gpu))
    e, eo = _rel(out, ref32), _rel(refbf, ref32)
    assert e < max(1.5e-2, 1.3 * eo), f"HIP vs f32 {e:.4e}, bf16 upstream vs f32 {eo:.4e}, ratio {e / max(eo, 1e-30):.2f} (bar 1.3)"


def test_clip_shape_encoder(gpu):
    from domain_rag_amd import vit
    from oracle import vit as ov
    cfg = vit.VitConfig(image_size=96, patch_size=32, hidden=128, heads=2, layers=2, intermediate=512, act=3, ln_eps=1e-5,
                        cls_token=True, patch_bias=False, proj_dim=64,
                        mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711))
    g = vit.init_generic_params(cfg, 2)
    img = (torch.rand(5, 96, 96, 3, generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    ref32 = ov.clip_image_embeds(g, 96, 32, 128, 2, 2, 512, 64, px, torch.float32)
    refbf = ov.clip_image_embeds(g, 96, 32, 128, 2, 2, 512, 64, px, torch.bfloat16)
    out = vit.VitHIP(cfg, g, gpu)(img.to(gpu))
    assert out.dtype == torch.float32 and out.shape == (5, 64)
    e, eo = _rel(out, ref32), _rel(refbf, ref32)
    assert e < max(1.5e-2, 1.3 * eo), f"HIP vs f32 {e:.4e}, bf16 upstream vs f32 {eo:.4e}, ratio {e / max(eo, 1e-30):.2f} (bar 1.3)"


@pytest.mark.parametrize("N,es,ps", [(1, [0.9], [1.0]), (2, [0.8, 1.0], [1.0, 1.0])])
def test_redux_prior(gpu, N, es, ps):
    from domain_rag_amd import redux, vit
    from oracle import redux as ored, vit as ov
    cfg = vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=2, intermediate=304)
    g = vit.init_generic_params(cfg, 3)
    rp = redux.init_redux_params(192, 256, seed=4)
    gen = torch.Generator().manual_seed(5)
    G = 2
    img = (torch.rand(G * N, 56, 56, 3, generator=gen) * 255).to(torch.uint8)
    t5 = torch.randn(24, 256, generator=gen).bfloat16()
    pooled = torch.randn(64, generator=gen).bfloat16()
    prior = redux.ReduxPriorHIP(cfg, g, rp, gpu)
    pe, pp = prior(img.to(gpu), t5.to(gpu), pooled.to(gpu), es, ps, group=N)
    assert pe.shape == (G, 24 + 16, 256) and pp.shape == (G, 64)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    lat32 = ov.siglip_last_hidden_state(g, 56, 14, 192, 2, 2, 304, px, torch.float32)
    rp32 = {k: v.float() for k, v in rp.items()}
    for gi in range(G):
        r_pe, r_pp = ored.redux_prior(lat32[gi * N:(gi + 1) * N], rp32, t5.float(), pooled.float(), es, ps)
        assert _rel(pe[gi], r_pe[0]) < 2e-2
        assert _rel(pp[gi], r_pp[0]) < 1e-2


def test_clip_embedding_is_independent_of_batch_history(gpu):
    """one image at a time (the reference's loop shape), then batches, then single again: every row must keep its bits
    (a B = 1 workspace once aliased the positional embedding and added the class token into it)"""
    from domain_rag_amd import retrieval as R
    model, _ = R.load_clip("ViT-B/32", device=gpu)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (45, 224, 224, 3), generator=g, dtype=torch.uint8).to(gpu)
    first = model.encode_image(x[:1]).clone()
    batch = model.encode_image(x).clone()           # M = 2250: the GEMMs mix both tile kernels
    again = model.encode_image(x[:1]).clone()
    small = model.encode_image(x[:7]).clone()
    assert torch.equal(first, again) and torch.equal(first[0], batch[0]) and torch.equal(small, batch[:7])


CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def test_clip_f32_tower_small_config_vs_transformers(gpu):
    """the float32 tower against upstream transformers' CLIPVisionModelWithProjection in float32 with shared weights;
    stated tolerance 2e-5 of the embedding scale (accumulation order only)"""
    from domain_rag_amd import vit
    from oracle import vit as ov
    cfg = vit.VitConfig(image_size=96, patch_size=32, hidden=128, heads=2, layers=2, intermediate=512, act=3, ln_eps=1e-5,
                        cls_token=True, patch_bias=False, proj_dim=64, mean=CLIP_MEAN, std=CLIP_STD)
    g = vit.init_generic_params(cfg, 2, dtype=torch.float32)
    img = (torch.rand(5, 96, 96, 3, generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    ref = ov.clip_image_embeds(g, 96, 32, 128, 2, 2, 512, 64, px, torch.float32)
    tower = vit.ClipVitF32HIP(cfg, g, gpu)
    out = tower(img.to(gpu))
    assert out.dtype == torch.float32 and out.shape == (5, 64)
    assert _rel(out, ref) < 2e-5, _rel(out, ref)
    # clip's `preprocess` output (normalised float NCHW, possibly on the CPU) is the other accepted input: same bits
    out2 = tower(px.to(gpu))
    assert torch.equal(out, out2)
    with pytest.raises(ValueError):
        vit.ClipVitF32HIP(vit.VitConfig.siglip_so400m(), g, gpu)


def test_clip_f32_vit_b32_vs_transformers_and_same_topk(gpu):
    """the real ViT-B/32 architecture (seeded weights): embeddings within 1e-4 of the float32 upstream model, and the
    ranking of a small corpus identical to the one computed from the upstream embeddings"""
    from domain_rag_amd import ops, retrieval as R, vit
    from oracle import retrieval as oret, vit as ov
    cfg = vit.VitConfig.clip_vit_b32()
    g = vit.init_generic_params(cfg, 7, dtype=torch.float32)
    gen = torch.Generator().manual_seed(3)
    base = torch.rand(40, 7, 7, 3, generator=gen)                      # low-frequency pictures, distinct per image
    img = (torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=224, mode="bilinear").permute(0, 2, 3, 1) * 255).to(torch.uint8)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    ref = ov.clip_image_embeds(g, 224, 32, 768, 12, 12, 3072, 512, px, torch.float32)
    model = R.ClipImageModel(vit.ClipVitF32HIP(cfg, g, gpu))
    out = model.encode_image(img.to(gpu))
    e = _rel(out, ref)
    assert e < 1e-4, e
    feats = ops.l2_normalize_(out.clone()).cpu().numpy()
    rfeats = (ref / ref.norm(dim=-1, keepdim=True)).numpy()
    D, I = oret.cosine_topk(feats[8:], feats[:8], 20)
    Dr, Ir = oret.cosine_topk(rfeats[8:], rfeats[:8], 20)
    import numpy as np
    assert np.array_equal(I, Ir) and np.abs(D - Dr).max() < 1e-5
    # the bf16 tower on the same weights sits two orders of magnitude further away (the deviation the fp32 default removes)
    eb = _rel(vit.VitHIP(cfg, g, gpu)(img.to(gpu)), ref)
    assert eb > 20 * e


def test_load_clip_precision_switch(gpu, monkeypatch):
    from domain_rag_amd import retrieval as R, vit
    m32, _ = R.load_clip("ViT-B/32", device=gpu)
    assert isinstance(m32.visual, vit.ClipVitF32HIP)
    mbf, _ = R.load_clip("ViT-B/32", device=gpu, precision="bf16")
    assert isinstance(mbf.visual, vit.VitHIP)
    monkeypatch.setenv("DRAG_CLIP_PRECISION", "bf16")
    assert isinstance(R.load_clip("ViT-B/32", device=gpu)[0].visual, vit.VitHIP)
    x = torch.randint(0, 256, (3, 224, 224, 3), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).to(gpu)
    a, b = m32.encode_image(x), mbf.encode_image(x)                      # same seeded weights: same embeddings up to bf16 error
    assert _rel(b, a) < 3e-2
    with pytest.raises(ValueError):
        R.load_clip("ViT-B/32", device=gpu, precision="fp16")


@pytest.mark.parametrize("B,T,H,hd", [(5, 10, 2, 64), (3, 50, 12, 64), (2, 64, 2, 32), (1, 1, 1, 64), (2, 17, 3, 48)])
def test_attention_small_f32_vs_torch(gpu, B, T, H, hd):
    import math
    from domain_rag_amd import ops
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, generator=torch.Generator().manual_seed(T)).to(gpu)
    out = torch.full((B * T, D + 4), 9.0, device=gpu)
    ops.attention_small_f32(qkv, out, B, T, H, hd, 3 * D, D + 4, 1 / math.sqrt(hd))
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(B, T, H, hd).permute(0, 2, 1, 3).double() for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    assert (out[:, :D].double() - ref).abs().max().item() < 2e-6 and torch.all(out[:, D:] == 9.0)
    with pytest.raises(RuntimeError, match="T <= 64"):
        ops.attention_small_f32(qkv, out, 1, 65, 1, 64, 192, 64, 1.0)
