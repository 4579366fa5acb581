"""persistent attention kernel (drag_set_option("attn_persist", 1)) against the product's one-item-per-workgroup kernel, interleaved"""
import math, os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
H = 24; D = H * 128
for (B, S) in [(8, 5337), (1, 5337)]:
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
    for rep in range(7):
        for per in (0, 1):
            ops.set_option("attn_persist", per)
            for name, fn in (("plain", plain), ("qprep", qprep)):
                if rep == 0: bench(fn, 2)
                t.setdefault((name, per), []).append(bench(fn))
    ops.set_option("attn_persist", 0)
    fl = 4.0 * S * S * 128 * H * B
    for k, v in sorted(t.items()):
        ms = statistics.median(v)
        print(f"B={B} S={S} {k[0]:5s} persist={'on' if k[1] else 'off'}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.0f} TFLOP/s", flush=True)
