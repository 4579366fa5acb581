"""oracle/ops_ref.py — per-kernel CPU references in plain torch (TEST INFRASTRUCTURE ONLY).
Each mirrors the torch/diffusers op the HIP kernel replaces, on bf16 tensors (torch CPU bf16 ops
accumulate in fp32 and round once per op, like the reference's CUDA bf16 pipeline)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

ACTS = {0: lambda x: x, 1: lambda x: F.gelu(x, approximate="tanh"), 2: F.silu,
        3: lambda x: x * torch.sigmoid(1.702 * x), 4: F.gelu}


def gemm_ref(a, w, bias=None, act=0, gate=None, resid=None, rows_per_batch=None, out_f32=False):
    """epi(a @ w.T): torch.nn.Linear + activation (+ gated residual  x + gate[b] * y)"""
    if out_f32:
        y = F.linear(a.float(), w.float(), None if bias is None else bias.float())
        return ACTS[act](y)
    y = ACTS[act](F.linear(a, w, bias))
    if gate is not None:
        B = gate.shape[0]
        y = (resid.view(B, rows_per_batch, -1) + gate[:, None] * y.view(B, rows_per_batch, -1)).view(y.shape)
    elif resid is not None:
        y = resid + y
    return y


def layernorm_modulate_ref(x, scale=None, shift=None, gamma=None, beta=None, eps=1e-6):
    """x [B, S, D]; AdaLN: LN(x) * (1 + scale[:, None]) + shift[:, None]; or affine nn.LayerNorm"""
    if gamma is not None:
        return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps)
    n = F.layer_norm(x, (x.shape[-1],), None, None, eps)
    if scale is None:
        return n
    return n * (1 + scale[:, None]) + shift[:, None]


def attention_ref(q, k, v, scale):
    """q,k,v [B, H, S, D] -> [B, S, H*D]  (F.scaled_dot_product_attention, no mask)"""
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False, scale=scale)
    return o.transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


def attention_ref_f64(q, k, v, scale):
    q, k, v = q.double(), k.double(), v.double()
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    return (p @ v).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)
