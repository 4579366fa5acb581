"""the product attention kernel and the 4-wave x 64-query experiment (DRAG_EXPERIMENTS build), 6 launches each at B = 8, S = 5337, for a --pmc pass"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H = 8, 5337, 24
D = H * 128
qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
s_pad = (S + 63) // 64 * 64
vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
for q64 in (2, 1):           # 2 = the 8-wave kernel, 1 = the 4-wave x 64-query kernel
    ops.set_option("attn_q64", q64)
    for _ in range(6):
        ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
    torch.cuda.synchronize()
ops.set_option("attn_q64", 0)
