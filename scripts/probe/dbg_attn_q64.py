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
run = lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
def bench():
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10
fl = 4.0 * S * S * 128 * H * B
for tune in (2, 6, 10, 14, 2):
    ops.set_option("attn_tune", tune)
    ms = min(bench() for _ in range(3))
    print(f"attn_tune={tune} (4: no barrier, 8: no vmcnt wait): {ms * 1e3:.0f} us = {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
ops.set_option("attn_tune", 2)
