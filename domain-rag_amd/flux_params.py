"""Parameter containers for the Flux DiT: config, synthetic (seeded) initialisation and checkpoint
loading.  Key names are the diffusers ``FluxTransformer2DModel`` state_dict names, so the weights the
reference loads with ``from_pretrained('./model/FLUX.1-dev' | 'FLUX.1-Fill-dev')``
(batch_generate_flux_kshot.py:148-151, outpainting_updown_sampling_redux.py:534-541) drop in unchanged.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import torch


@dataclass
class FluxConfig:
    in_channels: int = 64
    out_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: tuple = (16, 56, 56)
    mlp_ratio: int = 4

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @classmethod
    def flux_dev(cls) -> "FluxConfig":
        return cls()

    @classmethod
    def flux_fill(cls) -> "FluxConfig":
        return cls(in_channels=384)

    @classmethod
    def flux_schnell(cls) -> "FluxConfig":
        return cls(guidance_embeds=False)

    @classmethod
    def from_json(cls, path: str) -> "FluxConfig":
        with open(path) as f:
            c = json.load(f)
        return cls(in_channels=c.get("in_channels", 64), out_channels=c.get("out_channels") or 64,
                   num_layers=c.get("num_layers", 19), num_single_layers=c.get("num_single_layers", 38),
                   num_attention_heads=c.get("num_attention_heads", 24),
                   attention_head_dim=c.get("attention_head_dim", 128),
                   joint_attention_dim=c.get("joint_attention_dim", 4096),
                   pooled_projection_dim=c.get("pooled_projection_dim", 768),
                   guidance_embeds=c.get("guidance_embeds", True),
                   axes_dims_rope=tuple(c.get("axes_dims_rope", (16, 56, 56))))


def param_shapes(cfg: FluxConfig) -> dict[str, tuple]:
    D, J, P = cfg.dim, cfg.joint_attention_dim, cfg.pooled_projection_dim
    Hd = cfg.attention_head_dim
    F = cfg.mlp_ratio * D
    s: dict[str, tuple] = {}

    def lin(name, n, k, bias=True):
        s[name + ".weight"] = (n, k)
        if bias:
            s[name + ".bias"] = (n,)

    lin("x_embedder", D, cfg.in_channels)
    lin("context_embedder", D, J)
    emb = ["timestep_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else [])
    for e in emb:
        lin(f"time_text_embed.{e}.linear_1", D, 256)
        lin(f"time_text_embed.{e}.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, P)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        lin(p + "norm1.linear", 6 * D, D)
        lin(p + "norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[p + f"attn.{n}.weight"] = (Hd,)
        lin(p + "ff.net.0.proj", F, D)
        lin(p + "ff.net.2", D, F)
        lin(p + "ff_context.net.0.proj", F, D)
        lin(p + "ff_context.net.2", D, F)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}."
        lin(p + "norm.linear", 3 * D, D)
        lin(p + "proj_mlp", F, D)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k"):
            s[p + f"attn.{n}.weight"] = (Hd,)
        lin(p + "proj_out", D, D + F)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg.out_channels, D)
    return s


def init_params(cfg: FluxConfig, seed: int = 0, device: str | torch.device = "cpu",
                dtype: torch.dtype = torch.bfloat16, std: float = 0.02) -> dict[str, torch.Tensor]:
    """Seeded synthetic weights of the real architecture (there are no checkpoints offline).
    Linear weights ~ N(0, std), biases ~ N(0, std), RMSNorm scales ~ 1 + N(0, 0.1), modulation
    linears slightly larger so that gates/scales are exercised."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if "attn.norm" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        else:
            scale = std
            if name.endswith(".weight") and len(shape) == 2:
                # keep activations O(1) through depth: fan-in scaled
                scale = min(std * 2.5, 1.0 / (shape[1] ** 0.5))
            t = scale * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        out[name] = t.to(dtype)
    return out


def load_safetensors_dir(path: str, device="cpu", dtype=torch.bfloat16) -> dict[str, torch.Tensor]:
    """Load every ``*.safetensors`` shard of a diffusers transformer directory."""
    from safetensors.torch import load_file
    out: dict[str, torch.Tensor] = {}
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {path}")
    for f in files:
        for k, v in load_file(os.path.join(path, f), device=str(device)).items():
            out[k] = v.to(dtype)
    return out
