"""interleaved A/B of DRAG_GEMM_DBG = 0 / 1 (read per launch) on the headline's GEMM shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
VAR = os.environ.get("VAR", "1")
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(32768, 3072, 3072), (42696, 9216, 3072), (42696, 12288, 3072), (42696, 3072, 3072), (42696, 3072, 15360)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cr = torch.empty_like(C)
    best = {"0": 1e9, VAR: 1e9}
    for rnd in range(4):
        for d in ("0", VAR):
            os.environ["DRAG_GEMM_DBG"] = d
            best[d] = min(best[d], bench(lambda: ops.gemm(A, W, out=C)))
    os.environ["DRAG_GEMM_DBG"] = "0"; ops.gemm(A, W, out=Cr); os.environ["DRAG_GEMM_DBG"] = VAR; ops.gemm(A, W, out=C)
    tf = {k: 2 * M * N * K / v / 1e9 for k, v in best.items()}
    print(f"gemm {M}x{N}x{K}: base {tf['0']:.0f}  variant {tf[VAR]:.0f} TFLOP/s ({100 * (tf[VAR] / tf['0'] - 1):+.1f} %) same bits: {torch.equal(C, Cr)}", flush=True)
