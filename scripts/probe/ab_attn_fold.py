# needs scripts/probe/attention_fold_scale_max.patch applied (attn_tune bit 2 then selects the UNfolded 64-query kernel); the product build has no fold
"""the DiT's attention call (fused q preparation, B = 8, S = 5337): the 64-query kernel with the folded scale / maximum (attn_tune 2) against the
unfolded one (attn_tune 6 = 2 | 4) and the 8-wave kernel (attn_q64 2), interleaved in one process; output distances"""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H, s_txt = 8, 5337, 24, 1241
D = H * 128
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, S, 3 * D, device=dev, generator=g).bfloat16()
s_pad = (S + 63) // 64 * 64
vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
w = [(1 + 0.1 * torch.randn(128, device=dev, generator=g)).bfloat16() for _ in range(4)]
ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
ops.k_norm_rope_vt(qkv, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
run = lambda: ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128), w[0], w[2], cos, sin, s_txt)
VAR = {"8-wave": {"attn_q64": 2, "attn_tune": 2}, "q64": {"attn_q64": 0, "attn_tune": 6}, "q64 + fold": {"attn_q64": 0, "attn_tune": 2}}
def bench():
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10
t = {k: [] for k in VAR}
outs = {}
for rep in range(5):
    for k, opts in VAR.items():
        for n, v in opts.items(): ops.set_option(n, v)
        if rep == 0:
            run(); run(); torch.cuda.synchronize(); outs[k] = o.clone()
        t[k].append(bench())
ops.set_option("attn_q64", 0); ops.set_option("attn_tune", 2)
fl = 4.0 * S * S * 128 * H * B
for k in VAR:
    m = statistics.median(t[k])
    print(f"{k:12s} {m * 1e3:.0f} us = {fl / m / 1e9:.0f} TFLOP/s")
ref = outs["8-wave"].double()
for k in ("q64", "q64 + fold"):
    a = outs[k].double()
    print(f"{k} vs 8-wave: max |a - b| / max |b| = {((a - ref).abs().max() / ref.abs().max()).item():.3e}, equal elements {(a == ref).double().mean().item():.4f}")
# fp64 reference on a few heads for both
