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
N, K = 3072, 3072
for M in (9728, 9928, 9984, 10240, 12288 + 200, 65536 + 200):
    g = torch.Generator(device=dev).manual_seed(M)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = {2: [], 3: []}
    for k in t:
        ops.set_option("gemm_kernel", k); bench(lambda: ops.gemm(A, W, out=C), 3)
    for rep in range(5):
        for k in t:
            ops.set_option("gemm_kernel", k); t[k].append(bench(lambda: ops.gemm(A, W, out=C)))
    ops.set_option("gemm_kernel", 0)
    tiles = (M + 255) // 256 * 12
    print(f"M={M} ({tiles} tiles = {tiles / 256:.2f} rounds): 8-wave {statistics.median(t[2]) * 1e3:7.1f} us | 4-wave {statistics.median(t[3]) * 1e3:7.1f} us", flush=True)
