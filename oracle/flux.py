"""oracle/flux.py — CPU restatement (plain torch) of the Flux DiT denoiser the reference drives.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker; never by the product path.

What it restates: ``FluxTransformer2DModel.forward`` and its blocks from diffusers 0.33.1
(requirements.txt:7), which is NOT vendored under /root/reference and is not installable here —
the algorithm below is restated from the published architecture and anchored on the reference's
call sites: ``pipe(...)`` batch_generate_flux_kshot.py:467-474 and ``pipe_fill(...)``
outpainting_updown_sampling_redux.py:1246-1257 (bf16: batch_...:49, outpainting_...:28).
PARITY UNPINNED: the reference ships no tests or golden vectors and diffusers cannot be imported
offline, so this restatement is validated only by review + the block-level cross-checks against
`transformers` modules in tests/ (SURVEY §8c).

Parameter names follow the diffusers state_dict so a real checkpoint loads unchanged.
Run with dtype=torch.bfloat16 to mirror the reference's numerics op-for-op (torch CPU bf16
kernels accumulate in fp32), or dtype=torch.float32 for a high-precision yardstick.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


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


# --------------------------------------------------------------------------- pieces
def timestep_proj(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0) -> fp32 [B, dim]"""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def rope_tables(ids: torch.Tensor, axes_dims=(16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed: per axis, freqs = pos * theta^(-2i/dim) in float64; returns cos, sin fp32 [S, 64]."""
    cos, sin = [], []
    pos = ids.double()
    for i, d in enumerate(axes_dims):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i], freqs)
        cos.append(ang.cos().float())
        sin.append(ang.sin().float())
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


def latent_image_ids(h: int, w: int) -> torch.Tensor:
    """FluxPipeline._prepare_latent_image_ids for an h x w grid of packed latents."""
    ids = torch.zeros(h, w, 3)
    ids[..., 1] = torch.arange(h)[:, None]
    ids[..., 2] = torch.arange(w)[None, :]
    return ids.reshape(h * w, 3)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """diffusers RMSNorm: variance in fp32, cast to the weight dtype, then * weight"""
    dt = x.dtype
    var = x.float().pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)  # fp32 result (bf16 * fp32 promotes)
    y = y.to(weight.dtype) * weight
    return y.to(dt)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B, H, S, 128]; cos/sin [S, 64] (one value per interleaved pair)"""
    c = cos.repeat_interleave(2, dim=1)[None, None]
    s = sin.repeat_interleave(2, dim=1)[None, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * c + rot.float() * s).to(x.dtype)


def sdpa(q, k, v):
    return F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)


def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def _heads(x, H):
    B, S, _ = x.shape
    return x.view(B, S, H, -1).transpose(1, 2)


def ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


# --------------------------------------------------------------------------- blocks
def adaln_zero(p, name, temb, x):
    """AdaLayerNormZero: Linear(SiLU(temb)) -> six chunks in the order (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp,
    gate_mlp); returns (LN(x) * (1 + scale_msa) + shift_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp).  Pinned against the
    importable upstream twin `transformers...qwen2_5_omni.Qwen2_5_OmniAdaLayerNormZero` (tests/test_oracle_upstream_twins.py)."""
    mod = _lin(F.silu(temb), p, name)
    sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = mod.chunk(6, dim=1)
    return ln(x) * (1 + sc_msa[:, None]) + sh_msa[:, None], g_msa, sh_mlp, sc_mlp, g_mlp


def gated_mlp_residual(p, ff, x, attn_out, g_msa, sh_mlp, sc_mlp, g_mlp):
    """the half of a double-stream block behind its attention: x += gate_msa * attn; x += gate_mlp * FF(LN(x) * (1 + scale_mlp)
    + shift_mlp) with FF = Linear -> GELU(tanh) -> Linear.  Twin: qwen2_5_omni `DiTDecoderLayer.forward` behind `self.attn`."""
    x = x + g_msa[:, None] * attn_out
    n2 = ln(x) * (1 + sc_mlp[:, None]) + sh_mlp[:, None]
    h = _lin(F.gelu(_lin(n2, p, ff + "net.0.proj"), approximate="tanh"), p, ff + "net.2")
    return x + g_mlp[:, None] * h


def adaln_continuous(p, name, temb, x):
    """AdaLayerNormContinuous (norm_out): Linear(SiLU(temb)) -> (scale, shift) in THAT order.  Twin: qwen2_5_omni
    `Qwen2_5_OmniAdaLayerNormZero_Final`."""
    mod = _lin(F.silu(temb), p, name)
    scale, shift = mod.chunk(2, dim=1)
    return ln(x) * (1 + scale)[:, None] + shift[:, None]


def double_block(p, pre, cfg, hs, ehs, temb, cos, sin):
    H = cfg.num_attention_heads
    n_hs, g_msa, sh_mlp, sc_mlp, g_mlp = adaln_zero(p, pre + "norm1.linear", temb, hs)
    n_ehs, cg_msa, csh_mlp, csc_mlp, cg_mlp = adaln_zero(p, pre + "norm1_context.linear", temb, ehs)

    q = rms_norm(_heads(_lin(n_hs, p, pre + "attn.to_q"), H), p[pre + "attn.norm_q.weight"])
    k = rms_norm(_heads(_lin(n_hs, p, pre + "attn.to_k"), H), p[pre + "attn.norm_k.weight"])
    v = _heads(_lin(n_hs, p, pre + "attn.to_v"), H)
    eq = rms_norm(_heads(_lin(n_ehs, p, pre + "attn.add_q_proj"), H), p[pre + "attn.norm_added_q.weight"])
    ek = rms_norm(_heads(_lin(n_ehs, p, pre + "attn.add_k_proj"), H), p[pre + "attn.norm_added_k.weight"])
    ev = _heads(_lin(n_ehs, p, pre + "attn.add_v_proj"), H)
    q = apply_rope(torch.cat([eq, q], dim=2), cos, sin)
    k = apply_rope(torch.cat([ek, k], dim=2), cos, sin)
    v = torch.cat([ev, v], dim=2)
    o = sdpa(q, k, v).transpose(1, 2).reshape(hs.shape[0], -1, cfg.dim).to(q.dtype)
    St = ehs.shape[1]
    eo, o = o[:, :St], o[:, St:]
    o = _lin(o, p, pre + "attn.to_out.0")
    eo = _lin(eo, p, pre + "attn.to_add_out")

    hs = gated_mlp_residual(p, pre + "ff.", hs, o, g_msa, sh_mlp, sc_mlp, g_mlp)
    ehs = gated_mlp_residual(p, pre + "ff_context.", ehs, eo, cg_msa, csh_mlp, csc_mlp, cg_mlp)
    return ehs, hs


def single_block(p, pre, cfg, hs, temb, cos, sin):
    H = cfg.num_attention_heads
    mod = _lin(F.silu(temb), p, pre + "norm.linear")
    sh, sc, gate = mod.chunk(3, dim=1)
    n = ln(hs) * (1 + sc[:, None]) + sh[:, None]
    mlp = F.gelu(_lin(n, p, pre + "proj_mlp"), approximate="tanh")
    q = apply_rope(rms_norm(_heads(_lin(n, p, pre + "attn.to_q"), H), p[pre + "attn.norm_q.weight"]), cos, sin)
    k = apply_rope(rms_norm(_heads(_lin(n, p, pre + "attn.to_k"), H), p[pre + "attn.norm_k.weight"]), cos, sin)
    v = _heads(_lin(n, p, pre + "attn.to_v"), H)
    o = sdpa(q, k, v).transpose(1, 2).reshape(hs.shape[0], -1, cfg.dim).to(q.dtype)
    out = _lin(torch.cat([o, mlp], dim=2), p, pre + "proj_out")
    return hs + gate[:, None] * out


def time_text_embed(p, cfg, timestep, guidance, pooled):
    """CombinedTimestepGuidanceTextProjEmbeddings; timestep/guidance already scaled by 1000"""
    dt = pooled.dtype
    pre = "time_text_embed."
    te = _lin(F.silu(_lin(timestep_proj(timestep).to(dt), p, pre + "timestep_embedder.linear_1")), p,
              pre + "timestep_embedder.linear_2")
    if cfg.guidance_embeds:
        ge = _lin(F.silu(_lin(timestep_proj(guidance).to(dt), p, pre + "guidance_embedder.linear_1")), p,
                  pre + "guidance_embedder.linear_2")
        te = te + ge
    pe = _lin(F.silu(_lin(pooled, p, pre + "text_embedder.linear_1")), p, pre + "text_embedder.linear_2")
    return te + pe


def flux_forward(p: dict, cfg: FluxConfig, hidden, enc, pooled, timestep, img_ids, txt_ids, guidance=None,
                 taps: dict | None = None, time_dtype=None):
    """FluxTransformer2DModel.forward.  ``timestep`` is sigma in [0,1] (the pipeline passes t/1000);
    inside, timestep.to(dtype) * 1000 as diffusers does.  ``taps`` (optional dict) collects
    intermediate tensors for per-stage parity checks."""
    dt = hidden.dtype
    hs = _lin(hidden, p, "x_embedder")
    # diffusers: timestep.to(hidden_states.dtype) * 1000 — in the reference's bf16 pipeline the timestep
    # and guidance scale are therefore bf16-rounded (e.g. 3500 -> 3504).  ``time_dtype`` lets the fp32
    # yardstick use the same rounded times so that it differs from the bf16 run only by arithmetic.
    tdt = time_dtype or dt
    t = (timestep.to(tdt) * 1000).to(dt)
    g = (guidance.to(tdt) * 1000).to(dt) if guidance is not None else None
    temb = time_text_embed(p, cfg, t, g, pooled)
    ehs = _lin(enc, p, "context_embedder")
    cos, sin = rope_tables(torch.cat([txt_ids, img_ids], dim=0), cfg.axes_dims_rope)
    if taps is not None:
        taps["temb"] = temb
        taps["x_embed"] = hs
        taps["ctx_embed"] = ehs
    for i in range(cfg.num_layers):
        ehs, hs = double_block(p, f"transformer_blocks.{i}.", cfg, hs, ehs, temb, cos, sin)
        if taps is not None:
            taps[f"double.{i}"] = torch.cat([ehs, hs], dim=1)
    x = torch.cat([ehs, hs], dim=1)
    for i in range(cfg.num_single_layers):
        x = single_block(p, f"single_transformer_blocks.{i}.", cfg, x, temb, cos, sin)
        if taps is not None:
            taps[f"single.{i}"] = x
    x = x[:, enc.shape[1]:]
    return _lin(adaln_continuous(p, "norm_out.linear", temb, x), p, "proj_out")


# --------------------------------------------------------------------------- scheduler
def calculate_shift(seq_len, base_len=256, max_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_len - base_len)
    return seq_len * m + (base_shift - m * base_len)


def flow_sigmas(num_steps: int, seq_len: int):
    """FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=linspace(1, 1/n, n), mu=shift(seq_len)) with
    use_dynamic_shifting (FLUX.1-dev scheduler config): sigma' = e^mu / (e^mu + (1/sigma - 1)); then 0 appended.
    Returns (sigmas fp32 [n+1], timesteps = sigma' * 1000 fp32 [n])."""
    import numpy as np
    sig = np.linspace(1.0, 1.0 / num_steps, num_steps).astype(np.float32)      # the scheduler casts to float32 before shifting
    mu = calculate_shift(seq_len)
    sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)                  # float32 array against python floats: stays float32
    assert sig.dtype == np.float32
    sig = torch.from_numpy(sig)
    ts = sig * 1000.0
    return torch.cat([sig, torch.zeros(1)]), ts


def euler_step(x, v, sigma, sigma_next):
    """scheduler.step: prev = x.float() + (sigma_next - sigma) * v, cast back to x.dtype"""
    return (x.float() + (sigma_next - sigma) * v.float()).to(x.dtype)
