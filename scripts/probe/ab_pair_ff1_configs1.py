"""configs[1]'s ff up-projection pair (1024 + 512) x 12288 x 3072 + GELU: merged launch ("gemm_pair" 2) against the policy's two launches"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def tm(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M1, M2, N, K, act) in [(1024, 512, 12288, 3072, ops.ACT_GELU_TANH), (1024, 512, 3072, 3072, ops.ACT_NONE), (1024, 512, 9216, 3072, ops.ACT_NONE)]:
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda M: dict(a=torch.randn(M, K, device=dev, generator=g).bfloat16(), w=(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(),
                        out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), bias=torch.randn(N, device=dev, generator=g).bfloat16(), act=act)
    f, s2 = mk(M1), mk(M2)
    r = {}
    for v in (0, 2, 1):
        ops.set_option("gemm_pair", v)
        r[v] = statistics.median(tm(lambda: ops.gemm_pair(f, s2)) for _ in range(3))
    ops.set_option("gemm_pair", 0)
    print(f"({M1} + {M2}) x {N} x {K}: policy {r[0]:.1f} us | always merged {r[2]:.1f} us | never merged {r[1]:.1f} us")
