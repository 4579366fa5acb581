"""which KV tile of the 64-query attention kernel differs from the 8-wave kernel: V[j, j // 64] = 1 (tile mass in columns 0..31), V[j, 64 + j % 64] = 1"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, H, D = 1, 1, 128
for S in (1024, 1100, 1984):
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.zeros(B, S, 3 * D, device=dev)
    qkv[..., :D] = torch.randn(B, S, D, device=dev, generator=g)
    qkv[..., D:2 * D] = torch.randn(B, S, D, device=dev, generator=g)
    j = torch.arange(S, device=dev)
    v = torch.zeros(S, 128, device=dev)
    v[j, j // 64] = 1.0
    v[j, 64 + j % 64] = 1.0
    qkv[0, :, 2 * D:] = v
    qkv = qkv.bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
    outs = {}
    for q64 in (2, 1):
        ops.set_option("attn_q64", q64)
        o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        outs[q64] = o.float()[0]
    ops.set_option("attn_q64", 0)
    d = (outs[1] - outs[2]).abs()
    pert = d[:, :32].max(0).values
    print(f"S={S}: tiles with a difference (tile, max |diff| x S):", [(t, round(pert[t].item() * S, 2)) for t in range(32) if pert[t] > 1e-5])
    rows = (d[:, :32].max(1).values > 1e-5).nonzero().flatten().tolist()
    print("   rows mod 64 with a difference:", sorted(set(r % 64 for r in rows)))
