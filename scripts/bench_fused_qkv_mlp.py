"""why is the single-block to_q|k|v + proj_mlp as ONE launch not faster than two?  M = 42696, K = 3072, interleaved rounds:
separate (N = 9216 plain + N = 12288 GELU) vs fused N = 21504 in its variants (plain one destination / GELU from column 9216 / two destinations)"""
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
    return s.elapsed_time(e) / iters * 1e3
M, K, D, F = 42696, 3072, 3072, 12288
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(3 * D + F, K, device=dev) * 0.02).bfloat16(); b = torch.randn(3 * D + F, device=dev).bfloat16()
qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16); cat = torch.empty(M, D + F, device=dev, dtype=torch.bfloat16)
big = torch.empty(M, 3 * D + F, device=dev, dtype=torch.bfloat16)
V = {
    "separate qkv": lambda: ops.gemm(A, W[:3 * D], out=qkv, bias=b[:3 * D], M=M, lda=K, ldc=3 * D),
    "separate mlp (gelu)": lambda: ops.gemm(A, W[3 * D:], out=cat.view(-1)[D:], bias=b[3 * D:], act=ops.ACT_GELU_TANH, M=M, lda=K, ldc=D + F),
    "separate mlp (plain)": lambda: ops.gemm(A, W[3 * D:], out=cat.view(-1)[D:], bias=b[3 * D:], M=M, lda=K, ldc=D + F),
    "fused plain, one destination": lambda: ops.gemm(A, W, out=big, bias=b, M=M, lda=K, ldc=3 * D + F),
    "fused gelu from 9216, one destination": lambda: ops.gemm(A, W, out=big, bias=b, act=ops.ACT_GELU_TANH, act_n0=3 * D, M=M, lda=K, ldc=3 * D + F),
    "fused gelu from 9216, two destinations": lambda: ops.gemm(A, W, out=qkv, bias=b, act=ops.ACT_GELU_TANH, act_n0=3 * D, M=M, lda=K, ldc=3 * D,
                                                              out2=cat.view(-1)[D:], ldc2=D + F, n_split=3 * D),
}
def split_launches(w0, n, parts, act, out, ldc, col0):
    def run():
        step = n // parts
        for i in range(parts):
            ops.gemm(A, W[w0 + i * step: w0 + (i + 1) * step], out=out.view(-1)[col0 + i * step:], bias=b[w0 + i * step: w0 + (i + 1) * step], act=act, M=M, lda=K, ldc=ldc)
    return run
V["qkv as 3 launches of N = 3072"] = split_launches(0, 3 * D, 3, ops.ACT_NONE, qkv, 3 * D, 0)
V["mlp (gelu) as 2 launches of N = 6144"] = split_launches(3 * D, F, 2, ops.ACT_GELU_TANH, cat, D + F, D)
V["mlp (gelu) as 4 launches of N = 3072"] = split_launches(3 * D, F, 4, ops.ACT_GELU_TANH, cat, D + F, D)
t = {k: [] for k in V}
for rep in range(7):
    for k, fn in V.items():
        if rep == 0: bench(fn, 2)
        t[k].append(bench(fn))
med = {k: statistics.median(v) for k, v in t.items()}
for k, v in med.items():
    n = {"separate qkv": 3 * D, "separate mlp (gelu)": F, "separate mlp (plain)": F, "qkv as 3 launches of N = 3072": 3 * D,
         "mlp (gelu) as 2 launches of N = 6144": F, "mlp (gelu) as 4 launches of N = 3072": F}.get(k, 3 * D + F)
    print(f"{k:42s} {v:8.1f} us  {2 * M * n * K / v / 1e6:7.1f} TF/s")
print(f"separate total (gelu) {med['separate qkv'] + med['separate mlp (gelu)']:.1f} us; rounds: 23.48->24 + 31.31->32 = 56 vs fused 54.80->55")
