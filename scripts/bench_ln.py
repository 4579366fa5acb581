"""AdaLN LayerNorm at the DiT's shapes: the fixed-width kernel (all loads issued up front) vs the generic one, interleaved"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (B, S) in [(8, 5337), (8, 4096), (8, 1241), (1, 1536)]:
    D = 3072
    x = torch.randn(B, S, D, device=dev).bfloat16(); mod = (torch.randn(B, 6 * D, device=dev) * 0.3).bfloat16()
    y = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16)
    run = lambda: ops.layernorm(x, y, B * S, D, scale=mod.view(-1)[D:], shift=mod.view(-1), ldx=D, rows_per_batch=S, x_batch_stride=S * D, ld_mod=6 * D)
    t = {"generic": [], "fixed": []}
    for rep in range(5):
        for name, g in (("generic", 1), ("fixed", 0)):
            ops.set_option("ln_generic", g)
            if rep == 0: bench(run, 3)
            t[name].append(bench(run))
    ops.set_option("ln_generic", 0)
    gb = 2 * B * S * D * 2 / 1e9
    print(f"LN B={B} S={S}: " + " | ".join(f"{k} {statistics.median(v)*1e3:.1f} us = {gb/statistics.median(v):.2f} TB/s" for k, v in t.items()), flush=True)
