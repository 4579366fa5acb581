"""interleaved A/B of the specialised epilogue (drag_set_option "gemm_epilogue" 0) against the general one (1): plain, bias + GELU, gate + residual; bits must agree"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
B, S = 8, 5337
for (M, N, K, form) in [(42696, 9216, 3072, "bias"), (42696, 12288, 3072, "gelu"), (42696, 3072, 3072, "gate"), (42696, 3072, 12288, "gate"), (42696, 3072, 15360, "gate"), (32768, 3072, 3072, "plain")]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    gate = torch.randn(B, N, device=dev).bfloat16(); x0 = torch.randn(M, N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run():
        if form == "plain": ops.gemm(A, W, out=C)
        elif form == "bias": ops.gemm(A, W, out=C, bias=b)
        elif form == "gelu": ops.gemm(A, W, out=C, bias=b, act=ops.ACT_GELU_TANH)
        else: ops.gemm(A, W, out=C, bias=b, M=M, lda=K, ldc=N, c_rows_per_batch=S if M == B * S else M, c_batch_stride=S * N, gate=gate, resid=x0, ldg=N)
    VARS = tuple(os.environ.get("VARS", "1,0").split(","))
    best = {v: 1e9 for v in VARS}
    outs = {}
    for rnd in range(4):
        for d in VARS:
            ops.set_option("gemm_epilogue", int(d))
            best[d] = min(best[d], bench(run))
    for d in VARS:
        ops.set_option("gemm_epilogue", int(d))
        C.zero_(); run(); outs[d] = C.clone()
    tf = {k: 2 * M * N * K / v / 1e9 for k, v in best.items()}
    print(f"gemm {M}x{N}x{K} {form}: gemm_epilogue={VARS[0]}: {tf[VARS[0]]:.0f}  gemm_epilogue={VARS[1]}: {tf[VARS[1]]:.0f} TFLOP/s ({100 * (tf[VARS[1]] / tf[VARS[0]] - 1):+.1f} %) same bits: {torch.equal(outs[VARS[0]], outs[VARS[1]])}", flush=True)
    del A, W, C, x0
