"""round 5: fixed cost per launch / per tile round of the 4-wave kernel vs the 8-wave kernel: exactly 1, 2 and 4 rounds of 256 tiles at several K"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N) in [(4096, 4096), (8192, 4096), (16384, 4096)]:
    for K in (256, 1024, 3072):
        g = torch.Generator(device=dev).manual_seed(K)
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = {2: [], 3: []}
        for k in t:
            ops.set_option("gemm_kernel", k); bench(lambda: ops.gemm(A, W, out=C), 3)
        for rep in range(5):
            for k in t:
                ops.set_option("gemm_kernel", k); t[k].append(bench(lambda: ops.gemm(A, W, out=C)))
        ops.set_option("gemm_kernel", 0)
        print(f"{M // 256 * N // 256 // 256} round(s) of 256 tiles, K={K:5d} ({K // 64:3d} K-steps): 8-wave {statistics.median(t[2]) * 1e3:7.1f} us | 4-wave {statistics.median(t[3]) * 1e3:7.1f} us", flush=True)
