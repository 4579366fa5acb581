"""float32 matrix-core Linears of the CLIP ViT-B/32 tower (conv2d_f32_kernel as a 1x1 convolution) at batch 1024 / 256 / 100 tokens-rows:
TFLOP/s per shape.  DRAG_CONV_NO_LIN=1 takes the general convolution form (per-load predicates) for an A/B."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): fn()
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(51200, 3072, 768), (51200, 768, 3072), (51200, 2304, 768), (51200, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (5000, 2304, 768)]:
    x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.empty(M, N, device=dev)
    b = torch.randn(N, device=dev)
    t = statistics.median(bench(lambda: ops.linear_f32(x, w, y, M, ldx=K, ldy=N, bias=b)) for _ in range(3))
    print(f"f32 linear {M}x{N}x{K}: {2 * M * N * K / t / 1e9:.0f} TF/s ({t * 1e3:.0f} us)", flush=True)
