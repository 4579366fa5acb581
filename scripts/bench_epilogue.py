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
for (M, N, K) in [(42696, 12288, 3072), (42696, 9216, 3072), (32768, 3072, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.randn(M, N, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    gate = torch.randn(8, N, device=dev).bfloat16()
    rpb = M // 8
    outs = []
    for name, fn in [("plain", lambda: ops.gemm(A, W, out=C)), ("bias", lambda: ops.gemm(A, W, out=C, bias=b)),
                     ("bias+gelu", lambda: ops.gemm(A, W, out=C, bias=b, act=ops.ACT_GELU_TANH)),
                     ("bias+gate+resid", lambda: ops.gemm(A, W, out=C, bias=b, gate=gate, resid=C, ldg=N, c_rows_per_batch=rpb, c_batch_stride=rpb * N, ldc=N))]:
        ms = min(bench(fn) for _ in range(3))
        outs.append(f"{name} {2*M*N*K/ms/1e9:.0f}")
    print(f"{M}x{N}x{K}: " + " | ".join(outs), flush=True)
