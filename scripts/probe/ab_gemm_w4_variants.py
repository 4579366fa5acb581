"""round 5: schedule variants of gemm_bf16_w4 ("gemm_kernel" 400 + V) against the persistent 8-wave kernel (2): K-step slope from K = 3072 vs 12288.
V1 (no barrier) and V2 (no LDS-DMA in the loop) compute wrong values: timing ablations."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
KERNELS = [int(v) for v in os.environ.get("KERNELS", "2,400,401,402,403").split(",")]
M, N = 32768, 3072
res = {}
for K in (3072, 12288):
    g = torch.Generator(device=dev).manual_seed(K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = None
    t = {k: [] for k in KERNELS}
    same = {}
    for k in KERNELS:
        ops.set_option("gemm_kernel", k); bench(lambda: ops.gemm(A, W, out=C), 2)
        if ref is None: ref = C.clone()
        same[k] = bool(torch.equal(C, ref))
    for rep in range(5):
        for k in KERNELS:
            ops.set_option("gemm_kernel", k)
            t[k].append(bench(lambda: ops.gemm(A, W, out=C)))
    ops.set_option("gemm_kernel", 0)
    fl = 2 * M * N * K / 1e9
    for k in KERNELS: res[(k, K)] = statistics.median(t[k])
    print(f"K={K}: " + " | ".join(f"{k}: {fl / res[(k, K)]:.0f} TF/s{'' if same[k] else ' (bits differ)'}" for k in KERNELS), flush=True)
for k in KERNELS:
    print(f"   kernel {k}: {(res[(k, 12288)] - res[(k, 3072)]) * 1e3 / 144 / 6:.3f} us per K-step per tile round; per-tile constant {(res[(k, 3072)] * 1e3 / 6 - 48 * (res[(k, 12288)] - res[(k, 3072)]) * 1e3 / 144 / 6):.1f} us")
