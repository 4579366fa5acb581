"""what one workgroup of the attention kernel pays outside its KV loop (prologue: first K / V^T tiles + Q load (+ q RMSNorm / RoPE),
epilogue: O stores) — from launches whose workgroups walk 84 vs 168 KV tiles; and the 4-wave form (two workgroups per CU, whose seams
overlap each other's loops, at twice the LDS-DMA per flop)."""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
H = 24
D = H * 128
res = {}
for (B, S) in [(8, 5376), (2, 10752), (8, 5337)]:
    qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    wq = (1 + 0.1 * torch.randn(128, device=dev)).bfloat16()
    cos = torch.rand(S, 64, device=dev); sin = torch.rand(S, 64, device=dev)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    ops.k_norm_rope_vt(qkv, vt, wq, wq, cos, sin, B, S, H, 3 * D, 1241)
    sc = 1 / math.sqrt(128)
    plain = lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, sc)
    qprep = lambda: ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, sc, wq, wq, cos, sin, 1241)
    t = {}
    for rep in range(5):
        for w4 in (0, 1):
            ops.set_option("attn_w4", w4)
            for name, fn in (("plain", plain), ("qprep", qprep)):
                if rep == 0: bench(fn, 2)
                t.setdefault((name, w4), []).append(bench(fn))
    ops.set_option("attn_w4", 0)
    fl = 4.0 * S * S * 128 * H * B
    for k, v in t.items():
        ms = statistics.median(v)
        res[(B, S) + k] = ms
        print(f"B={B} S={S} {k[0]:5s} w4={k[1]}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.0f} TFLOP/s", flush=True)
# per-workgroup seam: T = rounds * (nkv * t_tile + seam);  (8, 5376): 21*24*8 = 4032 wgs = 15.75 rounds x 84 tiles; (2, 10752): 42*24*2 = 2016 wgs = 7.875 rounds x 168
for name in ("plain", "qprep"):
    T1, T2 = res[(8, 5376, name, 0)], res[(2, 10752, name, 0)]
    r1, r2 = 16.0, 8.0           # the longest CU runs ceil(15.75) and ceil(7.875) workgroups
    # T1 = r1 * (84 t + s), T2 = r2 * (168 t + s)
    a1, a2 = T1 / r1, T2 / r2
    t_tile = (a2 - a1) / 84
    seam = a1 - 84 * t_tile
    print(f"{name}: KV tile {t_tile*1e3:.3f} us, seam per workgroup {seam*1e3:.2f} us = {100*seam/a1:.1f} % of a 84-tile workgroup", flush=True)
