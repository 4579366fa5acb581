import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H = 1, 1024, 1
D = H * 128
for (sq, sk, sv) in [(1, 1, 6), (6, 1, 1), (1, 6, 1), (2, 2, 1), (3, 3, 1), (6, 6, 1), (6, 6, 6)]:
    g = torch.Generator().manual_seed(S + H)
    x = torch.randn(B, S, 3 * D, generator=g)
    x[..., :D] *= sq; x[..., D:2 * D] *= sk; x[..., 2 * D:] *= sv
    qkv = x.bfloat16().to(dev)
    vt = torch.empty(B, H, 128, (S + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
    ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
    outs = {}
    for q64 in (2, 1):
        ops.set_option("attn_q64", q64)
        o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        outs[q64] = o.float().cpu()[0]
    ops.set_option("attn_q64", 0)
    d = (outs[1] - outs[2]).abs()
    bad = (d.max(1).values > 0) | torch.isnan(outs[1]).any(1)
    nanrows = torch.isnan(outs[1]).any(1).nonzero().flatten().tolist()
    q = qkv[0, :, :D].float().cpu(); k = qkv[0, :, D:2 * D].float().cpu()
    s = (q @ k.T) / math.sqrt(128) * 1.4426950408889634
    # per row: in which 64-key tiles does the running maximum grow by more than 8 (after the first tile)?
    tm = s.view(S, S // 64, 64).max(2).values
    run = torch.cummax(tm, 1).values
    grow = (tm[:, 1:] - run[:, :-1]) > 8
    print(f"q x{sq} k x{sk} v x{sv}: rows differing {int(bad.sum())}/{S}, NaN rows {nanrows[:8]}; rows whose max grows by > 8 after tile 0: {int(grow.any(1).sum())}; "
          f"differing rows with no such growth in their 32-row group: {sum(1 for r in bad.nonzero().flatten().tolist() if not grow[(r // 32) * 32:(r // 32) * 32 + 32].any())}")
    for r in nanrows[:2]:
        print("   NaN row", r, "tile maxima (exp2 domain):", [round(v, 1) for v in tm[r].tolist()])
