"""round 5 probe: the 256x256 tile as 4 waves x (128 x 128) — gemm_bf16_deep<8, 2, 8> ("gemm_kernel" = 282; one wave per SIMD, a third less
LDS -> register traffic per flop than the 8-wave kernel) against the persistent 8-wave kernel (2).  The probe is NOT persistent (prologue
and epilogue in the open per tile), so the comparison that matters is the K-step slope: T(K2) - T(K1) over (K2 - K1) / 64 K-steps per tile round."""
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
KERNELS = [int(v) for v in os.environ.get("KERNELS", "2,400").split(",")]
for (M, N) in [(32768, 3072), (16384, 4096)]:
    res = {}
    for K in (3072, 6144, 12288):
        g = torch.Generator(device=dev).manual_seed(K)
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ref = None
        t = {k: [] for k in KERNELS}
        for k in KERNELS:
            ops.set_option("gemm_kernel", k); bench(lambda: ops.gemm(A, W, out=C), 2)
            if ref is None: ref = C.clone()
            else: print(f"   kernel {k} == kernel {KERNELS[0]} bit for bit: {torch.equal(C, ref)}")
        for rep in range(5):
            for k in KERNELS:
                ops.set_option("gemm_kernel", k)
                t[k].append(bench(lambda: ops.gemm(A, W, out=C)))
        ops.set_option("gemm_kernel", 0)
        fl = 2 * M * N * K / 1e9
        for k in KERNELS: res[(k, K)] = statistics.median(t[k])
        print(f"M={M} N={N} K={K}: " + " | ".join(f"kernel {k}: {res[(k, K)] * 1e3:.0f} us {fl / res[(k, K)]:.0f} TF/s" for k in KERNELS), flush=True)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = (tiles + 255) // 256
    for k in KERNELS:
        slope = (res[(k, 12288)] - res[(k, 3072)]) * 1e3 / ((12288 - 3072) / 64) / rounds
        print(f"   kernel {k}: {slope:.3f} us per K-step per tile round ({rounds} rounds)")
