"""tile policy at small M (BASELINE configs[1]: 1536 joint rows): 128x128 kernel vs the 256x256 one"""
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
for M in (1024, 1536, 1753):
    for (N, K) in [(3072, 3072), (9216, 3072), (12288, 3072), (21504, 3072), (3072, 12288), (3072, 15360)]:
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = {"t128": [], "t256": []}
        for rep in range(5):
            os.environ["DRAG_GEMM_T128"] = "1"; os.environ.pop("DRAG_GEMM_T256_MIN_M", None)
            if rep == 0: bench(lambda: ops.gemm(A, W, out=C), 3)
            t["t128"].append(bench(lambda: ops.gemm(A, W, out=C)))
            os.environ.pop("DRAG_GEMM_T128"); os.environ["DRAG_GEMM_T256_MIN_M"] = "1"
            if rep == 0: bench(lambda: ops.gemm(A, W, out=C), 3)
            t["t256"].append(bench(lambda: ops.gemm(A, W, out=C)))
        os.environ.pop("DRAG_GEMM_T256_MIN_M", None)
        fl = 2 * M * N * K / 1e9
        print(f"M={M} N={N} K={K} tiles256={((M+255)//256)*((N+255)//256)}: " + " | ".join(f"{k} {fl/statistics.median(v):.0f} TF/s" for k, v in t.items()), flush=True)
