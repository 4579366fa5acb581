"""FluxPriorReduxPipeline on the HIP path: SigLIP image encoder -> ReduxImageEncoder MLP ->
cat([T5 prompt embeds, image embeds]) -> per-image scales -> sum over the image axis.

Mirrors ``pipe_prior_redux(images, prompt=..., prompt_2=..., prompt_embeds_scale=[...],
pooled_prompt_embeds_scale=[...])`` (batch_generate_flux_kshot.py:459-465: two images, scales
[0.8, 1.0]; outpainting_updown_sampling_redux.py:1237-1243: one image).  The T5 / CLIP-text
embeddings of the (constant, per-dataset) prompt are an input: they are computed once and cached
by the caller, not on the per-image path.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .vit import VitConfig, VitHIP

T5_TOKENS, T5_DIM, POOLED_DIM = 512, 4096, 768


def param_shapes(siglip_dim: int = 1152, txt_dim: int = T5_DIM) -> dict:
    """tensor name -> shape of FLUX.1-Redux-dev/image_embedder (ReduxImageEncoder)"""
    mid = 3 * txt_dim
    return {"redux_up.weight": (mid, siglip_dim), "redux_up.bias": (mid,), "redux_down.weight": (txt_dim, mid), "redux_down.bias": (txt_dim,)}


def init_redux_params(siglip_dim: int = 1152, txt_dim: int = T5_DIM, seed: int = 0, device="cpu", dtype=torch.bfloat16):
    """ReduxImageEncoder: redux_up Linear(siglip_dim, 3*txt_dim), SiLU, redux_down Linear(3*txt_dim, txt_dim)"""
    g = torch.Generator(device=device).manual_seed(seed)
    mid = 3 * txt_dim

    def rn(*s, sc):
        return (sc * torch.randn(s, generator=g, device=device)).to(dtype)

    return {"redux_up.weight": rn(mid, siglip_dim, sc=1 / math.sqrt(siglip_dim)), "redux_up.bias": rn(mid, sc=0.02),
            "redux_down.weight": rn(txt_dim, mid, sc=1 / math.sqrt(mid)), "redux_down.bias": rn(txt_dim, sc=0.02)}


class ReduxPriorHIP:
    def __init__(self, vit_cfg: VitConfig, vit_params: dict, redux_params: dict, device="cuda"):
        self.dev = torch.device(device)
        self.vit = VitHIP(vit_cfg, vit_params, self.dev)
        self.w = {k: v.to(self.dev, torch.bfloat16).contiguous() for k, v in redux_params.items()}
        self.txt_dim = self.w["redux_down.weight"].shape[0]
        self._key = None

    def __call__(self, images_u8: torch.Tensor, t5_embeds: torch.Tensor, pooled: torch.Tensor, embeds_scale, pooled_scale,
                 group: int = 1):
        """images_u8 [G*N, S, S, 3] uint8 (already resized to the SigLIP size); every ``group``=N consecutive
        images form one prior call.  t5_embeds bf16 [Lt, txt_dim], pooled bf16 [P] (cached text encodings of the
        constant prompt).  embeds_scale / pooled_scale: N floats.  Returns prompt_embeds bf16
        [G, Lt + T, txt_dim] and pooled_prompt_embeds bf16 [G, P]."""
        n = images_u8.shape[0]
        N = group
        G = n // N
        T, Dv = self.vit.cfg.tokens, self.vit.cfg.hidden
        Lt, Dt, P = t5_embeds.shape[0], self.txt_dim, pooled.shape[-1]
        L = Lt + T
        if N < 1 or n % N != 0:
            raise ValueError(f"{n} images do not split into prior calls of group={N}")
        def per_image(v, what):       # N scales (one prior call's, repeated for every group) or n (one per image)
            v = [float(x) for x in v]
            if len(v) == n:
                return v
            if len(v) == N:
                return v * G
            raise ValueError(f"{what}: need {N} scales (per prior call) or {n} (per image), got {len(v)}")
        embeds_scale, pooled_scale = per_image(embeds_scale, "embeds_scale"), per_image(pooled_scale, "pooled_scale")
        key = (n, N, Lt, P, t5_embeds.data_ptr(), pooled.data_ptr())     # N: the output buffers are sized by G = n / N
        bf = dict(dtype=torch.bfloat16, device=self.dev)
        if self._key != key:
            slab = torch.empty((n, L, Dt), **bf)
            slab[:, :Lt].copy_(t5_embeds)                   # constant text rows, written once
            self._slab, self._mid = slab, torch.empty((n * T, self.w["redux_up.weight"].shape[0]), **bf)
            self._pooled_in = pooled.reshape(1, P).expand(n, P).contiguous()
            self._out, self._pout = torch.empty((G, L, Dt), **bf), torch.empty((G, P), **bf)
            self._key = key
        lat = self.vit(images_u8)                                                       # [n, T, Dv]
        ops.gemm(lat.view(-1, Dv), self.w["redux_up.weight"], out=self._mid, bias=self.w["redux_up.bias"], act=ops.ACT_SILU)
        ops.gemm(self._mid, self.w["redux_down.weight"], out=self._slab.view(-1)[Lt * Dt:], bias=self.w["redux_down.bias"],
                 M=n * T, lda=self._mid.shape[1], c_rows_per_batch=T, c_batch_stride=L * Dt, ldc=Dt)
        es = torch.tensor(embeds_scale, dtype=torch.float32).to(self.dev)
        ps = torch.tensor(pooled_scale, dtype=torch.float32).to(self.dev)
        ops.scale_sum(self._slab, es, self._out, G, N, L * Dt)
        ops.scale_sum(self._pooled_in, ps, self._pout, G, N, P)
        # fresh tensors: the work buffers are reused by the next call, and callers may keep several priors alive
        # (the diffusers pipeline returns new tensors too)
        return self._out.clone(), self._pout.clone()
