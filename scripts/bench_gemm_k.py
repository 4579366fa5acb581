"""GEMM time vs K at fixed M x N (fixed per-tile cost vs per-K-step cost), variants interleaved to cancel DVFS drift"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
VARIANTS = {"persistent + staged epilogue": {}, "fragment-layout epilogue": {"DRAG_GEMM_NARROW": "1"}, "one tile per workgroup": {"DRAG_GEMM_NONPERSISTENT": "1"}}
M, N = int(os.environ.get("M", 32768)), int(os.environ.get("N", 3072))
for K in (256, 1024, 3072, 12288):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = {k: [] for k in VARIANTS}
    for rep in range(7):
        for name, env in VARIANTS.items():
            os.environ.pop("DRAG_GEMM_NONPERSISTENT", None); os.environ.pop("DRAG_GEMM_NARROW", None); os.environ.update(env)
            if rep == 0: bench(lambda: ops.gemm(A, W, out=C), 3)
            t[name].append(bench(lambda: ops.gemm(A, W, out=C)))
    os.environ.pop("DRAG_GEMM_NONPERSISTENT", None); os.environ.pop("DRAG_GEMM_NARROW", None)
    lib = statistics.median(bench(lambda: torch.matmul(A, W.t(), out=C)) for _ in range(5))
    fl = 2 * M * N * K / 1e9
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    msg = " | ".join(f"{k} {statistics.median(v)*1e3:.0f} us ({fl/statistics.median(v):.0f} TF/s)" for k, v in t.items())
    print(f"K={K:6d} tiles={ntile}: {msg} | torch.matmul {lib*1e3:.0f} us ({fl/lib:.0f} TF/s)", flush=True)
