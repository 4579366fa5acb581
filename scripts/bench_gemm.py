import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (32768, 3072, 3072), (42696, 9216, 3072), (42696, 12288, 3072),
          (42696, 3072, 15360), (32768, 3072, 12288), (9928, 9216, 3072), (9928, 3072, 12288)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for env in ({}, {"DRAG_GEMM_NONPERSISTENT": "1"}):
        for k in ("DRAG_GEMM_NONPERSISTENT", "DRAG_GEMM_T128"): os.environ.pop(k, None)
        os.environ.update(env)
        ms = min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3))
        res.append(2 * M * N * K / ms / 1e9)
    for k in ("DRAG_GEMM_NONPERSISTENT", "DRAG_GEMM_T128"): os.environ.pop(k, None)
    print(f"gemm {M}x{N}x{K}: persistent {res[0]:.0f} | one tile per workgroup {res[1]:.0f} TF/s", flush=True)
    del A, W, C
