"""The metric's CHAINED denoise loop against the oracle, step by step: 30 Flux-Fill steps at strength 1.0
(outpainting_updown_sampling_redux.py:1246-1257; int(50 * 0.6) = 30 is the reference's Camouflage setting, :40) and the 50
txt2img steps of stage 2 (batch_generate_flux_kshot.py:467-474), identical seeds, at a size the CPU oracle affords: the real
width (D = 3072, 24 heads x 128, joint dim 4096, pooled 768, the full VAE) with 4 double + 8 single blocks (2 + 4 for the
50-step run) at 256 x 256 pixels.

Chained steps amplify rounding (SURVEY §7 "hard parts"), so the comparison is per step: after every Euler update the packed
latents of the HIP path and of the reference-dtype (bf16) oracle are both measured against the float32 oracle.  Bars, written
here: at EVERY step the HIP path's rms distance from float32 is at most 1.1 x the bf16 oracle's own rms distance (measured
1.00-1.011 at every one of the 30 + 50 steps: profiles/r03_chained_steps_*.json) and its max-norm distance at most 1.4 x the
oracle's (measured 1.00-1.27: the maximum over 16 384 latent elements of ONE realisation is a noisy statistic — two bf16
evaluations of one graph with different summation orders disagree on which element is the worst — so it gets the looser of
the two bars; the rms averages over all elements and is the stable measure).  Distances are relative to the float32 latents'
rms / max; a floor of 2e-3 covers the first steps, where both distances are a few bf16 ulps.  The error may not grow faster
than linearly in the step count; final pixels within max(1e-2, 1.3 x the bf16 oracle's distance) of full scale (measured 0.94 / 1.11).  The per-step
curves are printed and written to gpurun_out/ (the round's copy is committed under profiles/)."""
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

RATIO = 1.4          # max-norm
RATIO_RMS = 1.1
FLOOR = 2e-3


def _dist(a, ref):
    a, ref = a.double(), ref.double()
    d = (a - ref).abs()
    return (d.max() / ref.abs().max()).item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def _curves(hip_lat, taps32, tapsbf, steps):
    rows = []
    for i in steps:
        emax, erms = _dist(hip_lat[i], taps32[f"lat.{i}"])
        omax, orms = _dist(tapsbf[f"lat.{i}"], taps32[f"lat.{i}"])
        rows.append({"step": i, "hip_max": emax, "bf16_oracle_max": omax, "hip_rms": erms, "bf16_oracle_rms": orms,
                     "ratio_max": emax / max(omax, 1e-30), "ratio_rms": erms / max(orms, 1e-30)})
    return rows


def _check(rows, what):
    for r in rows:
        msg = (f"{what} step {r['step']}: HIP vs f32 max {r['hip_max']:.3e} rms {r['hip_rms']:.3e}; bf16 oracle vs f32 max "
               f"{r['bf16_oracle_max']:.3e} rms {r['bf16_oracle_rms']:.3e}; ratio max {r['ratio_max']:.2f} rms {r['ratio_rms']:.2f}")
        assert r["hip_max"] <= max(FLOOR, RATIO * r["bf16_oracle_max"]), msg
        assert r["hip_rms"] <= max(FLOOR / 4, RATIO_RMS * r["bf16_oracle_rms"]), msg
    # growth: from the first quarter on, the rms error grows at most linearly with the number of steps taken (a compounding
    # error would grow geometrically); 1.5 x slack for the step-to-step scatter
    n = len(rows)
    k = max(n // 4, 1)
    bound = 1.5 * (n / k) * max(rows[k - 1]["hip_rms"], FLOOR / 4)
    assert rows[-1]["hip_rms"] <= bound, f"{what}: rms error {rows[k - 1]['hip_rms']:.3e} after {k} steps -> {rows[-1]['hip_rms']:.3e} after {n}: super-linear"


def _report(name, rows, extra):
    print(f"[chained] {name}: step | HIP-f32 max  rms | bf16oracle-f32 max  rms | ratio max rms", flush=True)
    for r in rows:
        print(f"[chained] {name}: {r['step']:3d} | {r['hip_max']:.3e} {r['hip_rms']:.3e} | {r['bf16_oracle_max']:.3e} {r['bf16_oracle_rms']:.3e} | "
              f"{r['ratio_max']:.2f} {r['ratio_rms']:.2f}", flush=True)
    print(f"[chained] {name}: {json.dumps(extra)}", flush=True)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"chained_steps_{name}.json"), "w") as f:
            json.dump({"rows": rows, **extra}, f, indent=1)


def _setup(in_channels, layers, single_layers, seed, gpu, guidance_embeds=True):
    from domain_rag_amd import vae
    from domain_rag_amd.flux_params import FluxConfig, init_params
    from oracle import flux as oflux
    from conftest import oracle_threads
    oracle_threads(torch.bfloat16)
    cfg = FluxConfig(in_channels=in_channels, num_layers=layers, num_single_layers=single_layers, guidance_embeds=guidance_embeds)
    assert (cfg.dim, cfg.num_attention_heads, cfg.joint_attention_dim, cfg.pooled_projection_dim) == (3072, 24, 4096, 768)
    tp_dev = init_params(cfg, seed=seed, device=gpu)
    tp = {k: v.cpu() for k, v in tp_dev.items()}
    vcfg = vae.VaeConfig()
    vp = vae.init_params(vcfg, seed=seed + 1)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    return cfg, ocfg, tp_dev, tp, vcfg, vp


def _fill_chain(gpu, name, what, res, steps, St, layers, single_layers, seed, box, pixel_max_ratio=1.3, pixel_mean_ratio=None):
    from domain_rag_amd import fill_pipeline as fp, vae
    from domain_rag_amd.flux import FluxTransformerHIP
    from oracle import fill as ofill
    t_start = time.time()
    strength = 1.0
    cfg, ocfg, tp_dev, tp, vcfg, vp = _setup(384, layers, single_layers, seed, gpu)
    g = torch.Generator().manual_seed(seed + 1)
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    base = torch.stack([128 + 90 * torch.sin(xx / 23.0 + c) * torch.cos(yy / 17.0 - c) for c in range(3)], -1)
    image = (base + 8 * torch.randn(res, res, 3, generator=g)).clamp(0, 255).to(torch.uint8)[None]
    mask = torch.full((1, res, res), 255, dtype=torch.uint8); mask[:, box[0]:box[1], box[2]:box[3]] = 0
    pe = torch.randn(1, St, 4096, generator=g).bfloat16(); pp = torch.randn(1, 768, generator=g).bfloat16()
    en = torch.randn(1, 16, res // 8, res // 8, generator=g).bfloat16()
    mn = torch.randn(1, 16, res // 8, res // 8, generator=g).bfloat16()
    nt = torch.randn(1, (res // 16) ** 2, 64, generator=g).bfloat16()

    fill = fp.FluxFillHIP(FluxTransformerHIP(cfg, tp_dev, gpu), vae.FluxVaeHIP(vcfg, vp, gpu))
    hip_lat = {}
    out = fill(image.to(gpu), mask.to(gpu), pe.to(gpu), pp.to(gpu), guidance_scale=30.0, num_inference_steps=steps, strength=strength,
               enc_noise=en.to(gpu), masked_enc_noise=mn.to(gpu), noise_tokens=nt.to(gpu),
               on_step=lambda i, lat: hip_lat.__setitem__(i, lat.float().cpu())).cpu()
    del fill, tp_dev
    torch.cuda.empty_cache()
    assert sorted(hip_lat) == list(range(steps))
    taps, imgs = {}, {}
    from conftest import oracle_threads
    for oname, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        oracle_threads(dt)
        cast = (lambda d: {k: v.to(dt) for k, v in d.items()})
        taps[oname] = {}
        with torch.no_grad():
            _, img = ofill.fill_pipeline(cast(tp), ocfg, cast(vp), dict(block_out=vcfg.block_out_channels, layers=vcfg.layers_per_block),
                                         image, mask, pe, pp, 30.0, steps, strength, en, mn, nt, dtype=dt, taps=taps[oname])
        imgs[oname] = img.float()
    rows = _curves(hip_lat, taps["f32"], taps["bf16"], range(steps))
    hip = out.float() / 255.0
    e = (hip - imgs["f32"].permute(0, 2, 3, 1)).abs().max().item()
    e_or = (imgs["bf16"] - imgs["f32"]).abs().max().item()
    _report(name, rows, {"pipeline": what, "pixels_hip_vs_f32": e,
                         "pixels_bf16_oracle_vs_f32": e_or, "pixel_ratio": e / max(e_or, 1e-30), "seconds": time.time() - t_start})
    _check(rows, what)
    assert e <= max(1e-2 + 0.5 / 255, pixel_max_ratio * e_or), f"pixels: HIP vs f32 {e:.4f}, bf16 oracle vs f32 {e_or:.4f}, ratio {e / max(e_or, 1e-30):.2f}"
    if pixel_mean_ratio is not None:
        m = (hip - imgs["f32"].permute(0, 2, 3, 1)).abs().mean().item()
        m_or = (imgs["bf16"] - imgs["f32"]).abs().mean().item()
        assert m <= max(1e-3, pixel_mean_ratio * m_or), f"pixels, mean: HIP vs f32 {m:.5f}, bf16 oracle vs f32 {m_or:.5f}, ratio {m / max(m_or, 1e-30):.2f}"


def test_fill_30_chained_steps_vs_oracle_per_step(gpu):
    _fill_chain(gpu, "fill30", "Fill, 30 steps, strength 1.0, 256x256, 4 double + 8 single blocks at D=3072", 256, 30, 48, 4, 8, 20, (90, 166, 80, 170))


def test_fill_chained_steps_through_the_folded_64_query_attention(gpu):
    """round 6: the product's attention kernel at the headline's sequence lengths — attention_q64g_kernel with the scale fold — has its own
    rounding points (q c rounded once; -M in the score MFMAs' C operand), so the chained loop is taken through it as well: 512 x 512 pixels
    (1024 image + 100 text tokens = 18 KV tiles, the last one ragged), "attn_q64" = 1 (the policy takes the 64-query kernel from 4096 keys on),
    6 Fill steps over 2 double + 4 single blocks at the real width, the same per-step bars as the 30-step run"""
    from domain_rag_amd import _lib, ops
    try:
        ops.set_option("attn_q64", 1)
        assert _lib.load().drag_attention_bf16_choice(1024 + 100, 0, 1) == 641
        _fill_chain(gpu, "fill6_q64_fold", "Fill, 6 steps, strength 1.0, 512x512 (S = 1124), 2 double + 4 single blocks at D=3072, attention_q64g_kernel<true, true>",
                    512, 6, 100, 2, 4, 40, (180, 332, 160, 340), pixel_max_ratio=RATIO, pixel_mean_ratio=RATIO_RMS)
        # (pixels: the MAXIMUM over 786 432 values of one realisation moves by 15 % with the oracle's own summation order — 0.0203 with 64 host
        #  threads, 0.0177 with 32, HIP 0.0236 both times — so it gets this file's max-norm ratio, 1.4, and the MEAN the tight one, 1.1)
    finally:
        ops.set_option("attn_q64", 0)


def test_txt2img_50_chained_steps_vs_oracle_per_step(gpu):
    from domain_rag_amd import vae
    from domain_rag_amd.engine import FluxTxt2ImgHIP, generator_noise, pack_noise
    from domain_rag_amd.flux import FluxTransformerHIP
    from oracle import fill as ofill
    t_start = time.time()
    res, steps, St = 256, 50, 48
    cfg, ocfg, tp_dev, tp, vcfg, vp = _setup(64, 2, 4, 30, gpu)
    g = torch.Generator().manual_seed(31)
    pe = torch.randn(1, St, 4096, generator=g).bfloat16(); pp = torch.randn(1, 768, generator=g).bfloat16()
    noise = pack_noise(generator_noise(0, 1, res, res, 1)[0])          # stage 2's CPU generator, seed 0 (batch_...:52-61)
    pipe = FluxTxt2ImgHIP(FluxTransformerHIP(cfg, tp_dev, gpu), vae.FluxVaeHIP(vcfg, vp, gpu))
    hip_lat = {}
    out = pipe(pe.to(gpu), pp.to(gpu), height=res, width=res, guidance_scale=2.5, num_inference_steps=steps, noise_tokens=noise.to(gpu),
               on_step=lambda i, lat: hip_lat.__setitem__(i, lat.float().cpu())).cpu()
    del pipe, tp_dev
    torch.cuda.empty_cache()
    taps, imgs = {}, {}
    from conftest import oracle_threads
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        oracle_threads(dt)
        cast = (lambda d: {k: v.to(dt) for k, v in d.items()})
        taps[name] = {}
        with torch.no_grad():
            _, img = ofill.txt2img_pipeline(cast(tp), ocfg, cast(vp), dict(block_out=vcfg.block_out_channels, layers=vcfg.layers_per_block),
                                            pe, pp, 2.5, steps, res, res, noise, dtype=dt, taps=taps[name])
        imgs[name] = img.float()
    rows = _curves(hip_lat, taps["f32"], taps["bf16"], range(steps))
    hip = out.float() / 255.0
    e = (hip - imgs["f32"].permute(0, 2, 3, 1)).abs().max().item()
    e_or = (imgs["bf16"] - imgs["f32"]).abs().max().item()
    _report("txt2img50", rows, {"pipeline": "txt2img, 50 steps, guidance 2.5, 256x256, 2 double + 4 single blocks at D=3072", "pixels_hip_vs_f32": e,
                                "pixels_bf16_oracle_vs_f32": e_or, "pixel_ratio": e / max(e_or, 1e-30), "seconds": time.time() - t_start})
    _check(rows, "txt2img x50")
    assert e <= max(1e-2 + 0.5 / 255, 1.3 * e_or), f"pixels: HIP vs f32 {e:.4f}, bf16 oracle vs f32 {e_or:.4f}, ratio {e / max(e_or, 1e-30):.2f}"
