"""oracle/redux.py — CPU restatement of FluxPriorReduxPipeline.__call__ + ReduxImageEncoder
(diffusers 0.33.1, un-vendored; call sites batch_generate_flux_kshot.py:459-465 and
outpainting_updown_sampling_redux.py:1237-1243).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def redux_prior(image_latents, p, t5_embeds, pooled, embeds_scale, pooled_scale):
    """image_latents [N, T, Dv] (SigLIP last_hidden_state); t5_embeds [Lt, Dt]; pooled [P] ->
    prompt_embeds [1, Lt+T, Dt], pooled_prompt_embeds [1, P]"""
    dt = image_latents.dtype
    N = image_latents.shape[0]
    x = F.linear(F.silu(F.linear(image_latents, p["redux_up.weight"], p["redux_up.bias"])), p["redux_down.weight"],
                 p["redux_down.bias"])
    pe = torch.cat([t5_embeds[None].expand(N, -1, -1), x], dim=1)
    pp = pooled[None].expand(N, -1).clone()
    pe = pe * torch.tensor(embeds_scale, dtype=dt)[:, None, None]
    pp = pp * torch.tensor(pooled_scale, dtype=dt)[:, None]
    return pe.sum(dim=0, keepdim=True), pp.sum(dim=0, keepdim=True)
