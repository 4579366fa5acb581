"""tile-walk shape of the persistent 256x256 GEMM: M tiles per group (the 32 concurrent tiles of an XCD form a group_m x 32/group_m
super-tile; per K-step it fetches group_m A tiles + 32/group_m W tiles into that XCD's L2) — TFLOP/s on N(0,1) operands, interleaved"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
GROUPS = tuple(int(v) for v in os.environ.get("GROUPS", "2,4,6,8,12,16,32").split(","))
for (M, N, K) in [(42696, 9216, 3072), (42696, 12288, 3072), (42696, 3072, 15360), (32768, 3072, 3072), (32768, 3072, 12288)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = {g: [] for g in GROUPS}
    ref = None
    for rep in range(int(os.environ.get("REPS", "5"))):
        for g in GROUPS:
            ops.set_option("gemm_group_m", g)
            if rep == 0:
                bench(lambda: ops.gemm(A, W, out=C), 2)
                ref = C.clone() if ref is None else ref
                assert torch.equal(C, ref)
            t[g].append(bench(lambda: ops.gemm(A, W, out=C)))
    ops.set_option("gemm_group_m", 0)
    fl = 2 * M * N * K / 1e9
    print(f"M={M} N={N} K={K}: " + " | ".join(f"g{g} {fl/statistics.median(v):.0f}" for g, v in t.items()) + " TF/s", flush=True)
