import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (B, S, H) in [(1, 5337, 24), (8, 5337, 24), (8, 729, 16)]:
    D = H * 128
    qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    wq = torch.ones(128, device=dev).bfloat16()
    cos = torch.ones(S, 64, device=dev); sin = torch.zeros(S, 64, device=dev)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    ms_p = bench(lambda: ops.qk_norm_rope_vt(qkv, vt, wq, wq, wq, wq, cos, sin, B, S, H, 3 * D, 1241))
    fl = 4.0 * S * S * 128 * H * B
    res = []
    for abl in ("0",):
        os.environ["DRAG_ATTN_ABL"] = abl
        ms_a = min(bench(lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))) for _ in range(2))
        res.append(f"abl{abl} {fl/ms_a/1e9:.0f}")
    os.environ.pop("DRAG_ATTN_ABL")
    os.environ["DRAG_ATTN_W4"] = "1"
    ms_a = min(bench(lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))) for _ in range(2))
    res.append(f"4-wave {fl/ms_a/1e9:.0f}")
    os.environ.pop("DRAG_ATTN_W4")
    print(f"attn B={B} S={S} H={H}: prep {ms_p:.3f} ms | " + " | ".join(res) + " TF/s (0 real, 1 no-DMA, 2 no-exp, 3 no V ds_read)", flush=True)
