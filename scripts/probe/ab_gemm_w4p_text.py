"""round 5: why does the 4-wave kernel lose on the DiT's text-stream Linears (M = 9928 = 8 x 1241 rows)?  dense rows vs the joint buffer's row map"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, St, S, N, K = 8, 1241, 5337, 3072, 3072
M = B * St
g = torch.Generator(device=dev).manual_seed(0)
W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
bias = torch.randn(N, device=dev, generator=g).bfloat16(); gate = torch.randn(B, N, device=dev, generator=g).bfloat16()
Ad = torch.randn(M, K, device=dev, generator=g).bfloat16()
Aj = torch.randn(B, S, K, device=dev, generator=g).bfloat16()          # joint buffer: text rows are the first 1241 of each batch
Cd = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
Cj = torch.randn(B, S, N, device=dev, generator=g).bfloat16()
cases = {
    "dense A, dense C, plain": lambda: ops.gemm(Ad, W, out=Cd),
    "dense A, dense C, gate+resid": lambda: ops.gemm(Ad, W, out=Cd, bias=bias, gate=gate, resid=Cd, ldg=N, c_rows_per_batch=St, c_batch_stride=St * N),
    "joint A, dense C, plain": lambda: ops.gemm(Aj, W, out=Cd, M=M, lda=K, a_rows_per_batch=St, a_batch_stride=S * K),
    "dense A, joint C, gate+resid": lambda: ops.gemm(Ad, W, out=Cj, bias=bias, gate=gate, resid=Cj, ldg=N, M=M, lda=K, ldc=N, c_rows_per_batch=St, c_batch_stride=S * N),
    "joint A, joint C, gate+resid": lambda: ops.gemm(Aj, W, out=Cj, bias=bias, gate=gate, resid=Cj, ldg=N, M=M, lda=K, ldc=N, a_rows_per_batch=St, a_batch_stride=S * K,
                                                     c_rows_per_batch=St, c_batch_stride=S * N),
}
for name, fn in cases.items():
    t = {2: [], 3: []}
    for k in t:
        ops.set_option("gemm_kernel", k); bench(fn, 2)
    for rep in range(5):
        for k in t:
            ops.set_option("gemm_kernel", k); t[k].append(bench(fn))
    ops.set_option("gemm_kernel", 0)
    print(f"{name:32s}: 8-wave {statistics.median(t[2]) * 1e3:6.1f} us | 4-wave {statistics.median(t[3]) * 1e3:6.1f} us", flush=True)
