"""attention kernel variants, interleaved rounds in one process (DVFS drift cancels): KV-loop schedules, 8- vs 4-wave blocks"""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
# attn_q64: 2 = the 8-wave / 4-wave x 32-query family, 1 = the 4-wave x 64-query kernel (the policy's choice for S >= 4096)
VARIANTS = {"sched0": {"attn_sched": 0, "attn_q64": 2}, "sched2": {"attn_sched": 2, "attn_q64": 2}, "sched2+wide": {"attn_sched": 2, "attn_tune": 2, "attn_q64": 2},
            "q64+wide": {"attn_sched": 2, "attn_q64": 1, "attn_tune": 2}, "policy": {"attn_sched": 2, "attn_tune": 2, "attn_q64": 0}}
for (B, S, H) in [(1, 5337, 24), (8, 5337, 24), (8, 1753, 24), (8, 729, 16)]:
    D = H * 128
    qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    wq = torch.ones(128, device=dev).bfloat16()
    cos = torch.ones(S, 64, device=dev); sin = torch.zeros(S, 64, device=dev)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    ops.qk_norm_rope_vt(qkv, vt, wq, wq, wq, wq, cos, sin, B, S, H, 3 * D, 1241)
    fl = 4.0 * S * S * 128 * H * B
    run = lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
    run_v = lambda: ops.attention_v(qkv, qkv.view(-1)[D:], qkv.view(-1)[2 * D:], o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
    t = {k: [] for k in VARIANTS}
    t["row-major v"] = []
    for rep in range(6):
        for name, env in VARIANTS.items():
            for k in ("attn_sched", "attn_w4", "attn_tune", "attn_q64"): ops.set_option(k, 0)
            for k, v in env.items(): ops.set_option(k, v)
            if rep == 0: bench(run, 3)
            t[name].append(bench(run))
        ops.set_option("attn_sched", 2); ops.set_option("attn_w4", 0); ops.set_option("attn_tune", 2); ops.set_option("attn_q64", 0)
        if rep == 0: bench(run_v, 3)
        t["row-major v"].append(bench(run_v))
    ops.set_option("attn_sched", 2); ops.set_option("attn_w4", 0); ops.set_option("attn_tune", 2); ops.set_option("attn_q64", 0)
    print(f"attn B={B} S={S} H={H}: " + " | ".join(f"{k} {fl/statistics.median(v)/1e9:.0f} TF/s" for k, v in t.items()), flush=True)
