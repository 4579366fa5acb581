"""Flux-Fill composition (stage 3 of Domain-RAG) on the HIP path.

``FluxFillHIP.__call__`` mirrors ``pipe_fill(image=..., mask_image=..., height, width, guidance_scale,
num_inference_steps, prompt_embeds, pooled_prompt_embeds, generator, strength)`` and
``ReduxPriorHIP`` mirrors ``pipe_prior_redux([bg], ...)`` as ``process_sample_hires`` calls them
(outpainting_updown_sampling_redux.py:1237-1257).  ``SyntheticFillJob`` is bench.py's workload
(BASELINE configs[2]): seeded random weights of the real architectures and synthetic inputs of the
real shapes — there are no checkpoints or datasets offline.

Device-side data flow (no host round trips inside a batch):
  uint8 image/mask -> VAE encode x2 -> packed latents written straight into columns [0,64) / [64,128)
  of the DiT input rows, mask tokens into [128,384) -> scale_noise -> n x (DiT forward, strided Euler
  update of columns [0,64)) -> VAE decode reads those columns -> uint8 RGB.
"""
from __future__ import annotations

import torch

from . import ops, vae as vae_mod, vit as vit_mod, redux as redux_mod
from .flux import FluxTransformerHIP, latent_image_ids
from .flux_params import FluxConfig, init_params
from .scheduler import flow_sigmas, model_timestep, strength_start


class FluxFillHIP:
    """Transformer + VAE of FLUX.1-Fill-dev (in_channels 384) with the FluxFillPipeline call sequence."""

    def __init__(self, transformer: FluxTransformerHIP, vae: "vae_mod.FluxVaeHIP", use_graph: bool = True):
        self.tr, self.vae = transformer, vae
        self.dev = transformer.device
        self.use_graph = use_graph      # replay the DiT forward as a hipGraph (bit-identical outputs; frees the host thread)
        self._key = None

    def _buffers(self, B, H, W, St):
        key = (B, H, W, St)
        if self._key != key:
            h, w = H // 16, W // 16
            cfg = self.tr.cfg
            bf = dict(dtype=torch.bfloat16, device=self.dev)
            self._hidden = torch.empty((B, h * w, cfg.in_channels), **bf)
            # the prior's embeddings are copied here: stable addresses, so one captured graph serves every call of this shape
            self._pe, self._pp = torch.empty((B, St, cfg.joint_attention_dim), **bf), torch.empty((B, cfg.pooled_projection_dim), **bf)
            self._img_ids, self._txt_ids = latent_image_ids(h, w), torch.zeros(St, 3)
            self._key = key
        return self._hidden

    def __call__(self, image_u8: torch.Tensor, mask_u8: torch.Tensor, prompt_embeds: torch.Tensor, pooled: torch.Tensor, *,
                 guidance_scale: float, num_inference_steps: int, strength: float, enc_noise: torch.Tensor | None,
                 masked_enc_noise: torch.Tensor | None, noise_tokens: torch.Tensor, recorder=None, on_step=None) -> torch.Tensor:
        """image_u8 [B,H,W,3], mask_u8 [B,H,W] (255 = repaint), prompt_embeds bf16 [B,L,4096], pooled bf16 [B,768];
        enc_noise / masked_enc_noise bf16 [B,16,H/8,W/8] (generator draws for the two VAE posterior samples, None =
        mode); noise_tokens bf16 [B, n_tok, 64] (packed generator noise).  Returns uint8 RGB [B,H,W,3] on device."""
        B, H, W, _ = image_u8.shape
        h, w = H // 16, W // 16
        Si, St = h * w, prompt_embeds.shape[1]
        C = self.tr.cfg.in_channels
        hidden = self._buffers(B, H, W, St)
        self._pe.copy_(prompt_embeds); self._pp.copy_(pooled)
        prompt_embeds, pooled = self._pe, self._pp
        hv = hidden.view(-1)
        ops.set_recorder(recorder)
        try:
            # image latents -> columns [0,64); masked-image latents -> [64,128); mask -> [128,384)
            self.vae.encode_to_tokens(image_u8, None, enc_noise, hv, C)
            self.vae.encode_to_tokens(image_u8, mask_u8, masked_enc_noise, hv[64:], C)
            ops.mask_pack(mask_u8, hv[128:], B, H, W, C)
            sigmas, timesteps = flow_sigmas(num_inference_steps, Si)
            t0 = strength_start(num_inference_steps, strength)
            ops.scale_noise_rows(hv, noise_tokens, B * Si, 64, C, 64, float(sigmas[t0]))
            guidance = torch.full((B,), float(guidance_scale))
            for i in range(t0, num_inference_steps):
                t = torch.full((B,), model_timestep(timesteps[i]))
                timed = recorder is not None and (i - t0) % recorder.every == 0
                ops.set_recorder(recorder if timed else None)
                fwd = self.tr.forward_graphed if (self.use_graph and not timed) else self.tr.forward
                v = fwd(hidden, prompt_embeds, pooled, t, self._img_ids, self._txt_ids, guidance)
                ops.flow_euler_rows(hv, v, B * Si, 64, C, 64, float(sigmas[i + 1] - sigmas[i]))
                if on_step is not None:       # diffusers' callback_on_step_end sees the packed latents after the update
                    on_step(i, hidden[:, :, :64].clone())
            ops.set_recorder(recorder)
            return self.vae.decode_tokens(hv, B, h, w, ld=C).clone()      # the decoder's buffer is reused by the next call
        finally:
            ops.set_recorder(None)


class SyntheticFillJob:
    """bench.py workload: B composites at res x res, `denoise_steps` Flux-Fill steps, Redux prior included."""

    def __init__(self, batch: int = 8, res: int = 1024, denoise_steps: int = 30, device="cuda", seed: int = 0,
                 guidance: float = 30.0, strength: float = 1.0, image_prompt_scale: float = 1.0,
                 cfg: FluxConfig | None = None, vae_cfg=None, vit_cfg=None):
        self.B, self.res, self.n, self.dev = batch, res, denoise_steps, torch.device(device)
        self.cfg = cfg or FluxConfig.flux_fill()
        self.guidance, self.strength, self.ips = guidance, strength, image_prompt_scale
        dev = self.dev
        params = init_params(self.cfg, seed=seed, device=dev)
        tr = FluxTransformerHIP(self.cfg, params, dev)
        del params
        self.vae_cfg = vae_cfg or vae_mod.VaeConfig()
        vae = vae_mod.FluxVaeHIP(self.vae_cfg, vae_mod.init_params(self.vae_cfg, seed=seed + 1, device=dev), dev)
        self.fill = FluxFillHIP(tr, vae)
        self.vit_cfg = vit_cfg or vit_mod.VitConfig.siglip_so400m()
        self.prior = redux_mod.ReduxPriorHIP(self.vit_cfg, vit_mod.init_generic_params(self.vit_cfg, seed + 2, device=dev),
                                             redux_mod.init_redux_params(self.vit_cfg.hidden, self.cfg.joint_attention_dim,
                                                                         seed=seed + 3, device=dev), dev)
        g = torch.Generator(device=dev).manual_seed(seed + 4)
        S = self.vit_cfg.image_size
        # synthetic inputs: original image, keep-box mask (255 = repaint, 0 = keep a centred 300x300 box), background
        self.image = torch.randint(0, 256, (batch, res, res, 3), generator=g, device=dev, dtype=torch.uint8)
        self.mask = torch.full((batch, res, res), 255, device=dev, dtype=torch.uint8)
        c0, c1 = res // 2 - min(150, res // 4), res // 2 + min(150, res // 4)
        self.mask[:, c0:c1 + 1, c0:c1 + 1] = 0
        self.bg = torch.randint(0, 256, (batch, S, S, 3), generator=g, device=dev, dtype=torch.uint8)   # already SigLIP-sized
        # cached text encodings of the constant prompt "" (T5-XXL 512 tokens, CLIP-L pooled): synthetic stand-ins
        self.t5 = torch.randn((redux_mod.T5_TOKENS, self.cfg.joint_attention_dim), generator=g, device=dev).to(torch.bfloat16)
        self.pooled = torch.randn((self.cfg.pooled_projection_dim,), generator=g, device=dev).to(torch.bfloat16)
        lat = res // 8
        self.enc_noise = torch.randn((batch, 16, lat, lat), generator=g, device=dev).to(torch.bfloat16)
        self.menc_noise = torch.randn((batch, 16, lat, lat), generator=g, device=dev).to(torch.bfloat16)
        self.noise_tokens = torch.randn((batch, (res // 16) ** 2, 64), generator=g, device=dev).to(torch.bfloat16)
        self.Si, self.St = (res // 16) ** 2, redux_mod.T5_TOKENS + self.vit_cfg.tokens

    def stages(self):
        return ["redux_prior(SigLIP-so400m + Redux MLP + scale/sum)", "vae_encode(image)", "vae_encode(masked image)",
                "mask_pack", "scale_noise", f"denoise({self.n - strength_start(self.n, self.strength)}x Flux-Fill DiT + flow-Euler)",
                "vae_decode", "postprocess(uint8)"]

    def flops_per_image(self) -> float:
        """algorithmic FLOPs (2 x MACs) per composite: DiT (SURVEY §8d formula) + VAE enc x2 + dec + SigLIP/Redux"""
        cfg = self.cfg
        D, S = cfg.dim, self.St + self.Si
        blocks = cfg.num_layers + cfg.num_single_layers
        fwd = blocks * (2 * 12 * D * D * S + 2 * 2 * S * S * D)
        steps = self.n - strength_start(self.n, self.strength)
        scale = (self.res / 1024.0) ** 2
        vae = (10.5e12 + 2 * 5.0e12) * scale
        return float(fwd * steps + vae + 0.76e12)

    def run_batch(self, recorder=None) -> torch.Tensor:
        ops.set_recorder(recorder)
        try:
            pe, pp = self.prior(self.bg, self.t5, self.pooled, [self.ips], [1.0], group=1)
        finally:
            ops.set_recorder(None)
        return self.fill(self.image, self.mask, pe, pp, guidance_scale=self.guidance, num_inference_steps=self.n,
                         strength=self.strength, enc_noise=self.enc_noise, masked_enc_noise=self.menc_noise,
                         noise_tokens=self.noise_tokens, recorder=recorder)
