"""oracle/fill.py — CPU restatement of FluxFillPipeline.__call__ (diffusers 0.33.1, un-vendored) as
``process_sample_hires`` drives it (outpainting_updown_sampling_redux.py:1246-1257), composed from
oracle.vae / oracle.flux.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED."""
from __future__ import annotations

import torch

from . import flux as oflux
from . import vae as ovae


def fill_pipeline(tp, tcfg, vp, vae_kw, image_u8, mask_u8, prompt_embeds, pooled, guidance_scale, num_inference_steps,
                  strength, enc_noise, masked_enc_noise, noise_tokens, dtype=torch.bfloat16, taps=None):
    """returns (uint8 [B,H,W,3], float image in [0,1] NCHW).  Noise tensors stand for the generator draws:
    enc_noise / masked_enc_noise [B,16,H/8,W/8] for the two VAE posterior samples, noise_tokens [B,n,64] packed."""
    B, H, W, _ = image_u8.shape
    h, w = H // 16, W // 16
    init = ovae.preprocess_image(image_u8)                     # float32 [-1,1]
    mask = ovae.preprocess_mask(mask_u8)
    masked = init * (1 - mask)
    cast = (lambda t: t.to(dtype))
    img_lat = ovae.sample_latents(ovae.encode_moments(vp, cast(init), **vae_kw), None if enc_noise is None else cast(enc_noise))
    m_lat = ovae.sample_latents(ovae.encode_moments(vp, cast(masked), **vae_kw),
                                None if masked_enc_noise is None else cast(masked_enc_noise))
    sigmas, timesteps = oflux.flow_sigmas(num_inference_steps, h * w)
    init_t = min(num_inference_steps * strength, num_inference_steps)
    t0 = int(max(num_inference_steps - init_t, 0))
    sig0 = sigmas[t0].to(dtype)
    lat = ovae.pack_latents(img_lat)
    lat = sig0 * cast(noise_tokens) + (1.0 - sig0) * lat       # scheduler.scale_noise
    cond = torch.cat([ovae.pack_latents(m_lat), cast(ovae.pack_mask(mask))], dim=-1)
    img_ids, txt_ids = oflux.latent_image_ids(h, w), torch.zeros(prompt_embeds.shape[1], 3)
    guidance = torch.full((B,), float(guidance_scale))
    for i in range(t0, num_inference_steps):
        t = (timesteps[i].expand(B).to(torch.bfloat16) / 1000).float()      # pipeline: t.to(latents.dtype), then / 1000 in bf16
        v = oflux.flux_forward(tp, tcfg, torch.cat([lat, cond], dim=2), cast(prompt_embeds), cast(pooled), t, img_ids, txt_ids,
                               guidance, time_dtype=torch.bfloat16)
        lat = oflux.euler_step(lat, v, sigmas[i], sigmas[i + 1])
        if taps is not None:
            taps[f"lat.{i}"] = lat.clone()
    return ovae.decode_tokens_to_u8(vp, lat, h, w, **vae_kw)


def txt2img_pipeline(tp, tcfg, vp, vae_kw, prompt_embeds, pooled, guidance_scale, num_inference_steps, height, width,
                     noise_tokens, dtype=torch.bfloat16, taps=None):
    """FluxPipeline.__call__(prompt_embeds=..., pooled_prompt_embeds=..., guidance_scale, num_inference_steps, height,
    width, generator) as stage 2 calls it (batch_generate_flux_kshot.py:467-474); ``noise_tokens`` = packed generator draw."""
    B = prompt_embeds.shape[0]
    h, w = height // 16, width // 16
    sigmas, timesteps = oflux.flow_sigmas(num_inference_steps, h * w)
    lat = noise_tokens.to(dtype)
    img_ids, txt_ids = oflux.latent_image_ids(h, w), torch.zeros(prompt_embeds.shape[1], 3)
    guidance = torch.full((B,), float(guidance_scale)) if tcfg.guidance_embeds else None
    for i in range(num_inference_steps):
        t = (timesteps[i].expand(B).to(torch.bfloat16) / 1000).float()      # pipeline: t.to(latents.dtype), then / 1000 in bf16
        v = oflux.flux_forward(tp, tcfg, lat, prompt_embeds.to(dtype), pooled.to(dtype), t, img_ids, txt_ids, guidance,
                               time_dtype=torch.bfloat16)
        lat = oflux.euler_step(lat, v, sigmas[i], sigmas[i + 1])
        if taps is not None:
            taps[f"lat.{i}"] = lat.clone()
    return ovae.decode_tokens_to_u8(vp, lat, h, w, **vae_kw)
