import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H = 1, 1024, 1
D = 128
for (hot_tile, boost) in [(5, 40.0), (5, 300.0), (1, 300.0), (2, 300.0), (15, 300.0), (0, 300.0)]:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, S, 3 * D, generator=g) * 0.3
    qdir = torch.randn(D, generator=g); qdir /= qdir.norm()
    x[0, :, :D] += qdir * 4.0                       # every query has a component along qdir
    hot = hot_tile * 64 + 37
    x[0, hot, D:2 * D] = qdir * boost               # one key with a big score for every query: 4 * boost / sqrt(128) * log2e
    j = torch.arange(S)
    v = torch.zeros(S, 128); v[j, j // 64] = 1.0; v[j, 64 + j % 64] = 1.0
    x[0, :, 2 * D:] = v
    qkv = x.bfloat16().to(dev)
    vt = torch.empty(B, H, 128, S, device=dev, dtype=torch.bfloat16)
    ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
    outs = {}
    for q64 in (2, 1):
        ops.set_option("attn_q64", q64)
        o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        outs[q64] = o.float().cpu()[0]
    ops.set_option("attn_q64", 0)
    print(f"hot key in tile {hot_tile}, score boost {4 * boost / math.sqrt(128) * 1.4427:.1f} (exp2 domain): rows differing {int(((outs[1] - outs[2]).abs().max(1).values > 0).sum())}, NaN rows {int(torch.isnan(outs[1]).any(1).sum())}")
    for r in (0,):
        print(f"   row {r}: tile masses family {[round(v, 3) for v in outs[2][r, :16].tolist()]}")
        print(f"   row {r}: tile masses q64    {[round(v, 3) for v in outs[1][r, :16].tolist()]}")
