"""round 5: the four waves of gemm_bf16_w4p enter the epilogue together and meet at the LDS and at the store path in every segment; wave w
sleeps w * s * 64 cycles first ("gemm_epilogue" 16 + s with scripts/probe/gemm_w4p_epilogue_stagger.patch applied): TFLOP/s per s"""
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
S = [int(v) for v in os.environ.get("S", "0,1,2,3,4").split(",")]
ops.set_option("gemm_kernel", 3)
for (M, N, K) in [(32768, 3072, 3072), (32768, 9216, 3072), (32768, 12288, 3072), (32768, 3072, 12288)]:
    g = torch.Generator(device=dev).manual_seed(K + N)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    gate = torch.randn(8, N, device=dev, generator=g).bfloat16(); resid = torch.randn(M, N, device=dev, generator=g).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    forms = {"plain": dict(), "bias+gelu": dict(bias=bias, act=ops.ACT_GELU_TANH),
             "gate+resid": dict(bias=bias, gate=gate, resid=resid, ldg=N, c_rows_per_batch=M // 8, c_batch_stride=(M // 8) * N)}
    for name, kw in forms.items():
        t = {s: [] for s in S}
        for rep in range(5):
            for s in S:
                ops.set_option("gemm_epilogue", 16 + s if s else 0)
                t[s].append(bench(lambda: ops.gemm(A, W, out=C, **kw)))
        ops.set_option("gemm_epilogue", 0)
        fl = 2 * M * N * K / 1e9
        print(f"M={M} N={N} K={K} {name}: " + " | ".join(f"s={s}: {fl / statistics.median(t[s]):.0f}" for s in S), flush=True)
ops.set_option("gemm_kernel", 0)
