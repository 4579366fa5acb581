"""Is the GEMM power-limited?  Same kernel, same shape, operands of different switching activity: zeros, a constant, small
integers, N(0,1).  A kernel bound by its instruction schedule runs all of them at the same speed; one bound by the power
limit speeds up as the data toggles fewer bits.  torch.matmul (hipBLASLt's assembly kernel) beside it as the yardstick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from domain_rag_amd import ops
dev = torch.device("cuda:0")


def bench(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


M, N, K = 32768, 3072, 12288
g = torch.Generator(device=dev).manual_seed(0)
kinds = {
    "zeros": lambda *s: torch.zeros(*s, device=dev),
    "ones": lambda *s: torch.ones(*s, device=dev),
    "small ints {-1,0,1}": lambda *s: torch.randint(-1, 2, s, generator=g, device=dev).float(),
    "N(0,1)": lambda *s: torch.randn(*s, generator=g, device=dev),
}
for name, mk in kinds.items():
    A, W = mk(M, K).bfloat16(), mk(N, K).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ours = min(bench(lambda: ops.gemm(A, W, out=C)) for _ in range(3))
    lib = min(bench(lambda: torch.matmul(A, W.t(), out=C)) for _ in range(3))
    f = 2 * M * N * K / 1e9
    print(f"{name:22s} this kernel {f / ours:7.0f} TF/s   hipBLASLt {f / lib:7.0f} TF/s", flush=True)
    del A, W, C
