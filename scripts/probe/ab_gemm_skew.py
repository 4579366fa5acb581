import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(42696, 3072, 3072), (170784, 3072, 3072), (42696, 9216, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    variants = ["0", str(16 + (2 << 8)), str(16 + (4 << 8)), str(16 + (8 << 8)), "4"]
    best = {v: 1e9 for v in variants}
    for rnd in range(3):
        for v in variants:
            os.environ["DRAG_GEMM_DBG"] = v
            best[v] = min(best[v], bench(lambda: ops.gemm(A, W, out=C)))
    print(f"gemm {M}x{N}x{K}: " + "  ".join(f"dbg={v}: {best[v]*1e3:.0f} us" for v in variants), flush=True)
