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
N = 32768
for K in (256, 3072):
    for M in (512, 456, 385, 300):
        g = torch.Generator(device=dev).manual_seed(M)
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out = []
        for k, epi in ((2, 0), (3, 0), (3, 1)):
            ops.set_option("gemm_kernel", k); ops.set_option("gemm_epilogue", epi)
            bench(lambda: ops.gemm(A, W, out=C), 3)
            out.append(min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3)) * 1e3)
        ops.set_option("gemm_kernel", 0); ops.set_option("gemm_epilogue", 0)
        print(f"K={K} M={M} (128 full tiles + 128 {'ragged' if M < 512 else 'full'} ones): 8-wave {out[0]:6.1f} us | 4-wave {out[1]:6.1f} us | 4-wave, general epilogue {out[2]:6.1f} us", flush=True)
