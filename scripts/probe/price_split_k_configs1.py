"""round 5: what a split-K form of configs[1]'s single-block linear2 (1536 x 3072 x 15 360: 72 tiles of 256 x 256, or 256 of 96 x 192 with half the
arithmetic intensity) would cost: the three K-thirds as one launch of gemm_bf16_w4p over (3 x 1536) x 3072 x 5120 with f32 partials (216 tiles, one per
CU) + a pass that adds the partials and applies bias / gate / residual, against the launch the policy takes today"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N, K, S = 1536, 3072, 15360, 3
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
bias = torch.randn(N, device=dev, generator=g).bfloat16(); gate = torch.randn(1, N, device=dev, generator=g).bfloat16(); resid = torch.randn(M, N, device=dev, generator=g).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
kw = dict(bias=bias, gate=gate, resid=resid, ldg=N, c_rows_per_batch=M, c_batch_stride=M * N)
t_now = bench(lambda: ops.gemm(A, W, out=C, **kw))
print(f"today: kernel code {ops.gemm_choice(M, 0, N, K) if hasattr(ops, 'gemm_choice') else '?'}: {t_now:.1f} us = {2 * M * N * K / t_now / 1e6:.0f} TFLOP/s")
# the partial products: the same flops as one launch over stacked thirds (A2: [3 M, K / 3], W2: [N, K / 3]: W differs per third in the real thing — same traffic, same tiles)
A2 = torch.randn(S * M, K // S, device=dev, generator=g).bfloat16(); W2 = (torch.randn(N, K // S, device=dev, generator=g) * 0.02).bfloat16()
P = torch.empty(S * M, N, device=dev, dtype=torch.float32)
for code in (3, 2):
    ops.set_option("gemm_kernel", code)
    t = bench(lambda: ops.gemm(A2, W2, out=P))
    print(f"partials, kernel {code}, f32 out: {t:.1f} us")
    Pb = torch.empty(S * M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: ops.gemm(A2, W2, out=Pb))
    print(f"partials, kernel {code}, bf16 out (what the f32 epilogue costs): {t:.1f} us")
ops.set_option("gemm_kernel", 0)
P3 = P.view(S, M, N)
def reduce():
    y = (P3[0] + P3[1] + P3[2] + bias.float()).bfloat16()
    return (resid.float() + (gate.float() * y.float()).bfloat16().float()).bfloat16()
print(f"reduce pass as five torch kernels (upper bound; one fused pass moves {(S * 4 + 2 + 2) * M * N / 1e6:.0f} MB = {(S * 4 + 2 + 2) * M * N / 5e6:.1f} us at 5 TB/s): {bench(reduce):.1f} us")
