"""Flow-match Euler schedule as FluxPipeline / FluxFillPipeline set it up (diffusers 0.33.1,
un-vendored; reached from batch_generate_flux_kshot.py:467-474 and
outpainting_updown_sampling_redux.py:1246-1257).  Host-side scalar math only."""
from __future__ import annotations

import math

import numpy as np


def calculate_shift(image_seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.15) -> float:
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flow_sigmas(num_inference_steps: int, image_seq_len: int, dynamic_shift: bool = True, shift: float = 1.0):
    """sigmas (float32 [n+1], last = 0) and timesteps (= sigma*1000, float32 [n]).
    FLUX.1-dev / Fill-dev scheduler config: use_dynamic_shifting=True ->
    sigma' = e^mu / (e^mu + (1/sigma - 1)); FLUX.1-schnell: static shift=1.0 (identity)."""
    # set_timesteps casts the pipeline's float64 linspace to float32 FIRST and shifts in float32 (numpy keeps float32
    # against python-float scalars), so the shifted sigmas are float32-evaluated, not float64 values rounded afterwards
    sig = np.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps).astype(np.float32)
    if dynamic_shift:
        mu = calculate_shift(image_seq_len)
        sig = (math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)).astype(np.float32)
    else:
        sig = (shift * sig / (1 + (shift - 1) * sig)).astype(np.float32)
    ts = (sig * np.float32(1000.0)).astype(np.float32)
    return np.concatenate([sig, np.zeros(1, np.float32)]), ts


def strength_start(num_inference_steps: int, strength: float) -> int:
    """FluxFillPipeline.get_timesteps: index of the first step that is actually run."""
    init_timestep = min(num_inference_steps * strength, num_inference_steps)
    return int(max(num_inference_steps - init_timestep, 0))


def model_timestep(t: float) -> float:
    """what the pipelines hand to the transformer: ``timestep = t.expand(B).to(latents.dtype)`` (bf16) and then
    ``timestep / 1000`` — a bf16 tensor divided by a scalar, so the quotient is rounded to bf16 again.  (The transformer
    multiplies by 1000 in bf16 once more.)  Rounding t itself first matters: bf16(t / 1000) differs from
    bf16(bf16(t) / 1000) for about a quarter of the timesteps of a 30 / 50-step schedule."""
    import torch
    return float(torch.tensor(float(t), dtype=torch.float32).to(torch.bfloat16) / 1000)
