"""Flux-Fill composition job on synthetic inputs (bench.py's workload; BASELINE configs[2]).

Follows ``process_sample_hires`` -> ``pipe_prior_redux`` + ``pipe_fill``
(outpainting_updown_sampling_redux.py:1237-1257) with seeded random weights and inputs of the real
shapes: there are no checkpoints or datasets offline.
"""
from __future__ import annotations

import torch

from . import ops
from .flux import FluxTransformerHIP, latent_image_ids
from .flux_params import FluxConfig, init_params
from .scheduler import flow_sigmas, strength_start


class SyntheticFillJob:
    def __init__(self, batch: int = 8, res: int = 1024, denoise_steps: int = 30, device="cuda", seed: int = 0,
                 guidance: float = 30.0, strength: float = 1.0, cfg: FluxConfig | None = None,
                 txt_tokens: int = 512 + 729):
        self.B, self.res, self.n, self.dev = batch, res, denoise_steps, torch.device(device)
        self.cfg = cfg or FluxConfig.flux_fill()
        self.guidance, self.strength = guidance, strength
        params = init_params(self.cfg, seed=seed, device=self.dev)
        self.model = FluxTransformerHIP(self.cfg, params, self.dev)
        del params
        self.h = self.w = res // 16
        self.Si, self.St = self.h * self.w, txt_tokens
        g = torch.Generator(device=self.dev).manual_seed(seed + 1)
        bf = dict(device=self.dev, dtype=torch.bfloat16)
        # stand-ins until the encoders land (stages() says what is really executed)
        self.prompt_embeds = torch.randn((batch, self.St, self.cfg.joint_attention_dim), generator=g, device=self.dev).to(torch.bfloat16)
        self.pooled = torch.randn((batch, self.cfg.pooled_projection_dim), generator=g, device=self.dev).to(torch.bfloat16)
        self.cond = torch.randn((batch, self.Si, self.cfg.in_channels - 64), generator=g, device=self.dev).to(torch.bfloat16)
        self.noise = torch.randn((batch, self.Si, 64), generator=g, device=self.dev).to(torch.bfloat16)
        self.img_ids, self.txt_ids = latent_image_ids(self.h, self.w), torch.zeros(self.St, 3)
        self.sigmas, self.timesteps = flow_sigmas(denoise_steps, self.Si)
        self.t_start = strength_start(denoise_steps, strength)
        self.hidden = torch.empty((batch, self.Si, self.cfg.in_channels), **bf)

    def stages(self):
        return ["denoise(30x Flux-Fill DiT + flow-Euler)"]

    def flops_per_image(self) -> float:
        cfg = self.cfg
        D, S = cfg.dim, self.St + self.Si
        per_tok = 2 * 12 * D * D                      # MACs*2 per token per block (qkv, out, mlp)
        attn = 2 * 2 * S * S * D                      # QK^T + PV
        blocks = cfg.num_layers + cfg.num_single_layers
        fwd = blocks * (per_tok * S + attn)
        return float(fwd * (self.n - self.t_start))

    def run_batch(self, recorder=None):
        ops.set_recorder(recorder)
        try:
            lat = self.noise.clone()
            guidance = torch.full((self.B,), self.guidance)
            for i in range(self.t_start, self.n):
                self.hidden[:, :, :64].copy_(lat)
                self.hidden[:, :, 64:].copy_(self.cond)
                t = torch.full((self.B,), float(self.timesteps[i]) / 1000.0)
                v = self.model(self.hidden, self.prompt_embeds, self.pooled, t, self.img_ids, self.txt_ids, guidance)
                ops.flow_euler_step(lat, v.reshape(self.B, self.Si, 64), float(self.sigmas[i + 1] - self.sigmas[i]))
            return lat
        finally:
            ops.set_recorder(None)
