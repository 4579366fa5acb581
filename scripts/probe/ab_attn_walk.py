"""round 5: attention_q64_kernel with one workgroup per CU walking its items ("attn_walk" 0) against one item per workgroup (2), interleaved:
the DiT's call (B = 8, S = 5337, 24 heads) with and without the fused q preparation, and the stage-3 size (B = 2, S = 17 625)"""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, S, H, s_txt) in [(8, 5337, 24, 1241), (2, 17625, 24, 1241)]:
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(S)
    qkv = torch.randn(B, S, 3 * D, device=dev, generator=g).bfloat16()
    w = [(1 + 0.1 * torch.randn(128, device=dev, generator=g)).bfloat16() for _ in range(4)]
    ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    fl = 4.0 * S * S * 128 * H * B
    for qprep in (True, False):
        if qprep:
            ops.k_norm_rope_vt(qkv, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
            call = lambda: ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, out, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128), w[0], w[2], cos, sin, s_txt)
        else:
            ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
            call = lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, out, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
        t = {0: [], 2: []}; res = {}
        for v in (2, 0):
            ops.set_option("attn_walk", v); call(); torch.cuda.synchronize(); res[v] = out.clone()
        for rep in range(5):
            for v in (2, 0):
                ops.set_option("attn_walk", v); t[v].append(bench(call))
        ops.set_option("attn_walk", 0)
        a, b = statistics.median(t[2]), statistics.median(t[0])
        print(f"B={B} S={S} H={H} q prep {qprep}: one item per workgroup {a:.0f} us ({fl / a / 1e6:.0f} TFLOP/s) | walking {b:.0f} us ({fl / b / 1e6:.0f} TFLOP/s) | {100 * (a / b - 1):+.1f} % | same bits: {bool(torch.equal(res[0], res[2]))}", flush=True)
