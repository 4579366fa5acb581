"""End-to-end Flux-Fill composition (Redux prior -> VAE enc x2 -> DiT steps -> VAE dec) vs the CPU oracle,
reduced depth/width, identical seeds.  Stated tolerance (BASELINE north_star): pixels within 1e-2 relative."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fill_pipeline_vs_oracle(gpu):
    from domain_rag_amd import fill_pipeline as fp, redux, vae, vit
    from domain_rag_amd.flux import FluxTransformerHIP
    from domain_rag_amd.flux_params import FluxConfig, init_params
    from oracle import fill as ofill, flux as oflux, redux as ored, vit as ovit

    B, res, steps = 2, 64, 4
    cfg = FluxConfig(in_channels=384, num_layers=1, num_single_layers=2, num_attention_heads=2, joint_attention_dim=256,
                     pooled_projection_dim=64)
    tp = init_params(cfg, seed=0)
    vcfg = vae.VaeConfig(layers_per_block=1)
    vp = vae.init_params(vcfg, seed=1)
    vitcfg = vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=2, intermediate=304)
    vitp = vit.init_generic_params(vitcfg, 2)
    rp = redux.init_redux_params(192, 256, seed=3)
    g = torch.Generator().manual_seed(4)
    image = torch.randint(0, 256, (B, res, res, 3), generator=g, dtype=torch.uint8)
    mask = torch.full((B, res, res), 255, dtype=torch.uint8); mask[:, 20:41, 16:50] = 0
    bg = torch.randint(0, 256, (B, 56, 56, 3), generator=g, dtype=torch.uint8)
    t5 = torch.randn(24, 256, generator=g).bfloat16(); pooled = torch.randn(64, generator=g).bfloat16()
    en = torch.randn(B, 16, res // 8, res // 8, generator=g).bfloat16()
    mn = torch.randn(B, 16, res // 8, res // 8, generator=g).bfloat16()
    nt = torch.randn(B, (res // 16) ** 2, 64, generator=g).bfloat16()

    prior = redux.ReduxPriorHIP(vitcfg, vitp, rp, gpu)
    fill = fp.FluxFillHIP(FluxTransformerHIP(cfg, tp, gpu), vae.FluxVaeHIP(vcfg, vp, gpu))
    for strength in (1.0, 0.5):
        pe, pp = prior(bg.to(gpu), t5.to(gpu), pooled.to(gpu), [1.2], [1.0], group=1)
        out = fill(image.to(gpu), mask.to(gpu), pe, pp, guidance_scale=30.0, num_inference_steps=steps, strength=strength,
                   enc_noise=en.to(gpu), masked_enc_noise=mn.to(gpu), noise_tokens=nt.to(gpu)).cpu()
        # ---- oracle, fp32 yardstick and bf16 (reference dtype) runs
        res_or = {}
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            cast = (lambda d: {k: v.to(dt) for k, v in d.items()})
            lat = ovit.siglip_last_hidden_state(vitp, 56, 14, 192, 2, 2, 304, ovit.normalize_u8(bg, vitcfg.mean, vitcfg.std), dt)
            pes, pps = zip(*[ored.redux_prior(lat[i:i + 1], cast(rp), t5.to(dt), pooled.to(dt), [1.2], [1.0]) for i in range(B)])
            ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
            u8, img = ofill.fill_pipeline(cast(tp), ocfg, cast(vp), dict(block_out=vcfg.block_out_channels, layers=1), image, mask,
                                          torch.cat(pes), torch.cat(pps), 30.0, steps, strength, en, mn, nt, dtype=dt)
            res_or[name] = (u8, img.float())
        ref_u8, ref_img = res_or["f32"]
        e_or = (res_or["bf16"][1] - ref_img).abs().max().item()      # bf16 oracle vs fp32 oracle, full scale = 1
        e = (out.float() / 255.0 - ref_img.permute(0, 2, 3, 1)).abs().max().item()
        assert out.shape == (B, res, res, 3) and out.dtype == torch.uint8
        assert e < max(1e-2 + 0.5 / 255, 1.3 * e_or), f"strength {strength}: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"


@pytest.mark.parametrize("guidance_embeds,steps", [(True, 3), (False, 4)])
def test_txt2img_pipeline_vs_oracle(gpu, guidance_embeds, steps):
    """stage 2 (FLUX.1-dev shape, Redux over TWO images with scales [0.8, 1.0]) and the schnell-shape config
    (no guidance embedding, 4 steps) — BASELINE configs[1]"""
    from domain_rag_amd import redux, vae, vit
    from domain_rag_amd.engine import FluxTxt2ImgHIP, generator_noise, pack_noise
    from domain_rag_amd.flux import FluxTransformerHIP
    from domain_rag_amd.flux_params import FluxConfig, init_params
    from oracle import fill as ofill, flux as oflux, redux as ored, vit as ovit
    cfg = FluxConfig(in_channels=64, num_layers=2, num_single_layers=1, num_attention_heads=2, joint_attention_dim=256,
                     pooled_projection_dim=64, guidance_embeds=guidance_embeds)
    tp = init_params(cfg, seed=5)
    vcfg = vae.VaeConfig(layers_per_block=1)
    vp = vae.init_params(vcfg, seed=6)
    vitcfg = vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=2, intermediate=304)
    vitp = vit.init_generic_params(vitcfg, 7)
    rp = redux.init_redux_params(192, 256, seed=8)
    g = torch.Generator().manual_seed(9)
    imgs = torch.randint(0, 256, (2, 56, 56, 3), generator=g, dtype=torch.uint8)        # [retrieved, target]
    t5 = torch.randn(24, 256, generator=g).bfloat16(); pooled = torch.randn(64, generator=g).bfloat16()
    res = 64
    noise = pack_noise(generator_noise(0, 1, res, res, 1)[0])
    prior = redux.ReduxPriorHIP(vitcfg, vitp, rp, gpu)
    pe, pp = prior(imgs.to(gpu), t5.to(gpu), pooled.to(gpu), [0.8, 1.0], [1.0, 1.0], group=2)
    pipe = FluxTxt2ImgHIP(FluxTransformerHIP(cfg, tp, gpu), vae.FluxVaeHIP(vcfg, vp, gpu))
    out = pipe(pe, pp, height=res, width=res, guidance_scale=2.5, num_inference_steps=steps, noise_tokens=noise.to(gpu)).cpu()
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    outs = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        cast = (lambda d: {k: v.to(dt) for k, v in d.items()})
        lat = ovit.siglip_last_hidden_state(vitp, 56, 14, 192, 2, 2, 304, ovit.normalize_u8(imgs, vitcfg.mean, vitcfg.std), dt)
        r_pe, r_pp = ored.redux_prior(lat, cast(rp), t5.to(dt), pooled.to(dt), [0.8, 1.0], [1.0, 1.0])
        _, img = ofill.txt2img_pipeline(cast(tp), ocfg, cast(vp), dict(block_out=vcfg.block_out_channels, layers=1), r_pe, r_pp,
                                        2.5, steps, res, res, noise, dtype=dt)
        outs[name] = img.float()
    e_or = (outs["bf16"] - outs["f32"]).abs().max().item()
    e = (out.float() / 255.0 - outs["f32"].permute(0, 2, 3, 1)).abs().max().item()
    assert out.shape == (1, res, res, 3)
    assert e < max(1e-2 + 0.5 / 255, 1.3 * e_or), f"HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"


def test_prior_outputs_do_not_alias(gpu):
    """two priors computed back to back must both stay valid (callers batch several backgrounds)"""
    from domain_rag_amd import redux, vit
    vitcfg = vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=1, intermediate=304)
    prior = redux.ReduxPriorHIP(vitcfg, vit.init_generic_params(vitcfg, 2), redux.init_redux_params(192, 256, seed=3), gpu)
    g = torch.Generator().manual_seed(1)
    bg = torch.randint(0, 256, (2, 56, 56, 3), generator=g, dtype=torch.uint8).to(gpu)
    t5 = torch.randn(24, 256, generator=g).bfloat16().to(gpu); pooled = torch.randn(64, generator=g).bfloat16().to(gpu)
    a, pa = prior(bg[0:1], t5, pooled, [1.0], [1.0], group=1)
    a0 = a.clone()
    b, pb = prior(bg[1:2], t5, pooled, [1.0], [1.0], group=1)
    assert torch.equal(a, a0) and not torch.equal(a, b) and a.data_ptr() != b.data_ptr() and pa.data_ptr() != pb.data_ptr()


def test_sampled_roofline_recorder_does_not_change_the_pixels(gpu):
    """bench.py brackets the GEMMs of every n-th denoise step with events and lets the other steps replay the hipGraph:
    the mixed eager/replay batch must produce the bytes of the plain product run, and count the launches it says."""
    from domain_rag_amd import fill_pipeline as fp, ops, vae, vit
    from domain_rag_amd.flux_params import FluxConfig
    cfg = FluxConfig(in_channels=384, num_layers=1, num_single_layers=2, num_attention_heads=2, joint_attention_dim=256,
                     pooled_projection_dim=64)
    job = fp.SyntheticFillJob(batch=2, res=64, denoise_steps=5, device=gpu, seed=7, cfg=cfg, vae_cfg=vae.VaeConfig(layers_per_block=1),
                              vit_cfg=vit.VitConfig(image_size=56, patch_size=14, hidden=192, heads=2, layers=2, intermediate=304))
    plain = job.run_batch().cpu()
    rec_all, rec_2 = ops.GemmRecorder(), ops.GemmRecorder(every=2)
    a = job.run_batch(recorder=rec_all).cpu()
    b = job.run_batch(recorder=rec_2).cpu()
    assert torch.equal(plain, a) and torch.equal(plain, b)
    assert torch.equal(plain, job.run_batch().cpu())
    n_all, n_2 = rec_all.totals()[2], rec_2.totals()[2]
    per_step = (n_all - n_2) // 2                      # steps 1 and 3 of 0..4 are not bracketed at every=2
    assert per_step > 0 and n_all - n_2 == 2 * per_step and (n_all - 5 * per_step) == (n_2 - 3 * per_step) > 0
