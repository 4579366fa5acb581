import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H, D = 1, 1024, 1, 128
ops.set_option("attn_q64", 1)
res = {}
for tile in (5, 6):
    bad = []
    for pos in range(64):
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, S, 3 * D, generator=g) * 0.3
        qdir = torch.randn(D, generator=g); qdir /= qdir.norm()
        x[0, :, :D] += qdir * 4.0
        x[0, tile * 64 + pos, D:2 * D] = qdir * 300.0
        qkv = x.bfloat16().to(dev)
        vt = torch.empty(B, H, 128, S, device=dev, dtype=torch.bfloat16)
        ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
        o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        nanrows = torch.isnan(o.float()[0]).any(1).nonzero().flatten().tolist()
        if nanrows:
            bad.append((pos, len(nanrows), sorted(set(r % 64 for r in nanrows))[:4], sorted(set(r % 64 for r in nanrows))[-1]))
    print(f"tile {tile}: hot-key positions with NaN rows (pos, rows, first row%64.., last):", bad)
ops.set_option("attn_q64", 0)
