"""Library yardstick (NOT on the product path): how fast do rocBLAS/hipBLASLt (torch.matmul) and torch SDPA run the
pipeline's shapes on this GPU, next to the hand-written kernels.  Tells which kernel has known headroom."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
shapes = [(8192, 8192, 8192), (32768, 9216, 3072), (32768, 3072, 3072), (32768, 12288, 3072), (32768, 3072, 12288),
          (42696, 21504, 3072), (42696, 3072, 15360), (9928, 9216, 3072), (9928, 3072, 12288)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ours = min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3))
    libt = min(bench(lambda: torch.matmul(A, W.t(), out=C)) for _ in range(3))
    fl = 2 * M * N * K / 1e9
    print(f"gemm {M}x{N}x{K}: ours {fl/ours:.0f} TF/s | torch.matmul {fl/libt:.0f} TF/s", flush=True)
    del A, W, C
for (B, S) in ((8, 5337), (1, 5337), (8, 1024)):
    H, D = 24, 128
    q = torch.randn(B, H, S, D, device=dev).bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q)
    try:
        ms = min(bench(lambda: F.scaled_dot_product_attention(q, k, v)) for _ in range(3))
        print(f"sdpa B={B} S={S}: torch {4*B*H*S*S*D/ms/1e9:.0f} TF/s ({ms:.2f} ms)", flush=True)
    except Exception as ex:
        print("sdpa failed:", ex)
