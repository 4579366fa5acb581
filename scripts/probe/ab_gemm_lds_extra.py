"""round 5 calibration: what does LDS -> register traffic cost gemm_bf16_t256 in clock / throughput?  "gemm_mfma" 6 | 12 run the same kernel
with 6 | 12 more ds_read_b128 per phase that nobody uses (+25 % | +50 % fragment reads per K-step: 24 -> 36 -> 48 of 1 KiB per wave).
Interleaved timing, then the effective clock from the run's own GRBM counter is taken in a separate --pmc pass (scripts/pmc_gemm_walk.sh MFMAS=0,6,12)."""
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
for (M, N, K) in [(32768, 9216, 3072), (32768, 3072, 12288), (42696, 21504, 3072)]:
    g = torch.Generator(device=dev).manual_seed(K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = {v: [] for v in (0, 6, 12)}
    ref = None
    for v in t:
        ops.set_option("gemm_mfma", v); bench(lambda: ops.gemm(A, W, out=C), 2)
        if ref is None: ref = C.clone()
        assert torch.equal(C, ref)
    for rep in range(7):
        for v in t:
            ops.set_option("gemm_mfma", v)
            t[v].append(bench(lambda: ops.gemm(A, W, out=C)))
    ops.set_option("gemm_mfma", 0)
    fl = 2 * M * N * K / 1e9
    print(f"M={M} N={N} K={K}: " + " | ".join(f"+{v} reads/phase {fl / statistics.median(ms):.0f} TF/s" for v, ms in t.items()), flush=True)
