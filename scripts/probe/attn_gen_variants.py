"""interleaved A/B of the generated attention stream's schedule variants (scripts/gen/attn_q64_tile.py VARIANTS; DRAG_EXPERIMENTS builds:
"attn_gen" = 10 + index) on the DiT's call (B = 8, S = 5337, 24 heads, fused q preparation).  Variants 1-5 move the LDS-DMA pieces (same
arithmetic: same bits as the default fold stream, checked); 6 / 7 are timing ablations (no staging / no barrier: wrong results)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from domain_rag_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, S, H, s_txt = 8, 5337, 24, 1241
D = H * 128
scale = 1 / math.sqrt(128)
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, S, 3 * D, generator=g).bfloat16().to(dev)
w = [(1 + 0.1 * torch.randn(128, generator=g)).bfloat16().to(dev) for _ in range(4)]
ang = torch.rand(S, 64, generator=g) * 6.28
cos, sin = torch.cos(ang).contiguous().to(dev), torch.sin(ang).contiguous().to(dev)
vt = torch.empty(B, H, 128, (S + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
ops.k_norm_rope_vt(qkv, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
flops = 4.0 * S * S * 128 * H * B
exp = ops.experiments_built()
gens = [1, 2, 0] + ([29] if exp else [])
names = {1: "hand-placed attention_q64_kernel", 2: "generated, no fold", 0: "generated + fold (product: pieces under the trailing P V)", 11: "fold, two pieces per step in steps 0..3",
         12: "fold, pieces in steps 4..11", 13: "fold, pieces under the trailing P V", 14: "fold, K at the top / V^T in steps 0..3", 15: "fold, K at the top / V^T in steps 4..7",
         16: "ABLATION no staging", 17: "ABLATION no barrier", 18: "ABLATION no fragment reads", 19: "ABLATION v_mov for v_exp", 20: "ABLATION no softmax VALU",
         21: "ABLATION no maxima trees", 22: "ABLATION MFMAs and waits only", 23: "fold, row sums by v_pk_add_f32", 24: "fold, fragment reads three steps ahead",
         25: "fold, v_pk_add_f32 + reads three steps ahead", 10: "fold, one piece per step in steps 0..7 (first product form)",
         26: "fold, row sums by v_dot2c_f32_bf16 of the packed P", 27: "fold, fragment reads ONE step ahead (rings of 2)",
         28: "fold, row sums by ones x P MFMAs (8 more MFMAs, 64 fewer adds per tile)", 29: "fold, first-key-half maxima in the previous tile's steps 9..15"}


def call(o):
    ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, scale, w[0], w[2], cos, sin, s_txt)


def timed(gen, reps=12):
    ops.set_option("attn_gen", gen)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        call(o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call(o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, o


try:
    res = {gg: [] for gg in gens}
    outs = {}
    for rnd in range(5):
        for gg in gens:
            us, o = timed(gg)
            res[gg].append(us)
            outs[gg] = o
    for gg in gens:
        us = sorted(res[gg])[len(res[gg]) // 2]
        same = ""
        if gg in (11, 12, 13, 14, 15, 24, 27, 29):
            same = "  same bits as the default fold stream" if torch.equal(outs[gg], outs[0]) else "  BITS DIFFER FROM THE DEFAULT FOLD STREAM"
        if gg == 28:
            d = (outs[28].float() - outs[0].float()).abs()
            same = f"  vs the default fold stream: max |diff| {d.max().item():.2e} of max {outs[0].float().abs().max().item():.2e}, {100 * (d == 0).float().mean().item():.1f} % equal"
        if gg == 2:
            same = "  same bits as the hand-placed kernel" if torch.equal(outs[2], outs[1]) else "  BITS DIFFER FROM THE HAND-PLACED KERNEL"
        print(f"attn_gen {gg:2d}  {names[gg]:58s} {[round(x) for x in res[gg]]} us  median {us:.0f} us = {flops / us / 1e6:.0f} TFLOP/s{same}", flush=True)
finally:
    ops.set_option("attn_gen", 0)
