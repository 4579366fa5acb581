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
    assert e < max(1.5e-2, 2.5 * eo), (e, eo)


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
    assert e < max(1.5e-2, 2.5 * eo), (e, eo)


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
