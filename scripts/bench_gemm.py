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
    for flag in (None, "1"):
        if flag: os.environ["DRAG_GEMM_T128"] = flag
        else: os.environ.pop("DRAG_GEMM_T128", None)
        ms = bench(lambda: ops.gemm(A, W, out=C))
        res.append(2 * M * N * K / ms / 1e9)
    os.environ.pop("DRAG_GEMM_T128", None)
    print(f"gemm {M}x{N}x{K}: t256 {res[0]:.1f} TF/s | t128 {res[1]:.1f} TF/s", flush=True)
    del A, W, C
