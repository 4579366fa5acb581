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
loops = [int(x) for x in os.environ.get("LOOPS", "1,0").split(",")]
for (M, N, K) in [(32768, 3072, 3072), (42696, 3072, 15360), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for loop in loops:
        ops.set_option("gemm_t256_loop", loop)
        ms = min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3))
        print(f"dbg={os.environ.get('DRAG_GEMM_DBG','0')} gemm {M}x{N}x{K} loop={loop}: {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
