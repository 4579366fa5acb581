"""which key positions / output columns of the 64-query attention kernel differ from the 8-wave kernel: q = 0 (uniform softmax) with V rows that
name their key (V[j, d] = 1 iff j % 64 == d % 64), then random q with the same V; prints the per-(key slot) error of one head"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, H = 1, 1
D = H * 128
for S, qscale in ((1024, 0.0), (1024, 1.0), (1088, 1.0)):
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.zeros(B, S, 3 * D, device=dev)
    qkv[..., :D] = torch.randn(B, S, D, device=dev, generator=g) * qscale
    qkv[..., D:2 * D] = torch.randn(B, S, D, device=dev, generator=g)
    j = torch.arange(S, device=dev)
    v = torch.zeros(S, 128, device=dev)
    v[j, j % 64] = 1.0
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
    print(f"S={S} qscale={qscale}: max diff {d.max().item():.4f} (ref max {outs[2].abs().max().item():.4f}); rows with a difference {int((d.max(1).values > 1e-4).sum())}/{S}")
    percol = d.max(0).values
    bad = [(c, round(percol[c].item(), 4)) for c in range(128) if percol[c] > 1e-4]
    print("   columns (= key slot within a 64-key tile, twice):", bad[:64])
    if qscale == 0.0:
        print("   row 0 of the 64-query kernel x S:", [round(x, 2) for x in (outs[1][0, :64] * S).tolist()])
    if qscale == 1.0 and S == 1024:
        badrows = (d.max(1).values > 1e-4).nonzero().flatten().tolist()
        print("   bad rows mod 256:", sorted(set(r % 256 for r in badrows))[:80])
        r = badrows[0]
        print(f"   row {r}: slot, ref x S, got x S")
        print("   ", [(c, round(outs[2][r, c].item() * S, 2), round(outs[1][r, c].item() * S, 2)) for c in range(0, 64)])
