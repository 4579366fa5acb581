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
for (M, N, K) in [(8192, 8192, 8192), (42696, 9216, 3072), (42696, 3072, 15360)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out = []
    for abl in ("0", "1", "2", "3", "4"):
        os.environ["DRAG_GEMM_ABL"] = abl
        ms = min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3))
        out.append(f"abl{abl} {2*M*N*K/ms/1e9:.0f}")
    os.environ.pop("DRAG_GEMM_ABL")
    print(f"gemm {M}x{N}x{K}: " + " | ".join(out) + "   (0 real, 1 no-DMA, 2 no-ds_read, 3 neither, 4 DMA without waits)", flush=True)
