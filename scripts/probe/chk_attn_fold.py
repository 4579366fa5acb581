# needs scripts/probe/attention_fold_scale_max.patch applied (attn_tune bit 2 then selects the UNfolded 64-query kernel); the product build has no fold
import math, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H, s_txt = 1, 4300, 2, 100
D = H * 128
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, S, 3 * D, device=dev, generator=g).bfloat16()
s_pad = (S + 63) // 64 * 64
vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
w = [(1 + 0.1 * torch.randn(128, device=dev, generator=g)).bfloat16() for _ in range(4)]
ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
ops.k_norm_rope_vt(qkv, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
outs = {}
for name, opts in {"q64_policy_first": {"attn_q64": 0, "attn_tune": 2}, "8w": {"attn_q64": 2, "attn_tune": 2}, "q64u": {"attn_q64": 0, "attn_tune": 6}, "q64u_t4": {"attn_q64": 0, "attn_tune": 4}, "q64f": {"attn_q64": 0, "attn_tune": 2}}.items():
    for n, v in opts.items(): ops.set_option(n, v)
    o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128), w[0], w[2], cos, sin, s_txt)
    torch.cuda.synchronize()
    outs[name] = o.float().cpu()
    print(name, "nan:", torch.isnan(outs[name]).sum().item(), "rows with nan:", torch.isnan(outs[name]).any(-1).sum().item())
r = outs["8w"]
for k in outs:
    print(k, "equal", (outs[k] == r).float().mean().item(), "maxrel", ((outs[k] - r).abs().max() / r.abs().max()).item())
