"""round 5: the persistent 4-wave GEMM ("gemm_kernel" 410, experiment build) against the persistent 8-wave kernel (2) on the DiT's launch shapes:
bit equality (plain, bias + GELU, gate + residual on the DiT's row map) and interleaved TFLOP/s."""
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
KERNELS = [int(v) for v in os.environ.get("KERNELS", "2,410").split(",")]
SHAPES = [(32768, 3072, 3072), (32768, 9216, 3072), (32768, 12288, 3072), (32768, 3072, 12288), (42696, 21504, 3072), (42696, 3072, 15360), (2500, 1024, 1024)]
for (M, N, K) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(K + N)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    B = 8 if M % 8 == 0 else 1
    gate = torch.randn(B, N, device=dev, generator=g).bfloat16(); resid = torch.randn(M, N, device=dev, generator=g).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    forms = {"plain": dict(), "bias+gelu": dict(bias=bias, act=ops.ACT_GELU_TANH),
             "gate+resid": dict(bias=bias, gate=gate, resid=resid, ldg=N, c_rows_per_batch=M // B, c_batch_stride=(M // B) * N)}
    line = []
    for name, kw in forms.items():
        ref, t, same = None, {k: [] for k in KERNELS}, {}
        for k in KERNELS:
            ops.set_option("gemm_kernel", k); bench(lambda: ops.gemm(A, W, out=C, **kw), 2)
            if ref is None: ref = C.clone()
            same[k] = bool(torch.equal(C, ref))
        for rep in range(5):
            for k in KERNELS:
                ops.set_option("gemm_kernel", k)
                t[k].append(bench(lambda: ops.gemm(A, W, out=C, **kw)))
        ops.set_option("gemm_kernel", 0)
        fl = 2 * M * N * K / 1e9
        line.append(f"{name}: " + " vs ".join(f"{fl / statistics.median(t[k]):.0f}{'' if same[k] else ' (BITS DIFFER)'}" for k in KERNELS))
    print(f"M={M} N={N} K={K}: " + " | ".join(line) + f"   TFLOP/s, kernels {KERNELS}", flush=True)
