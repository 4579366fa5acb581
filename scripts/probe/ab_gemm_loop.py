"""A/B of the 256x256 GEMM's two K loops in ONE process, interleaved (drag_set_option "gemm_t256_loop": 0 = the hand-placed symmetric loop
(t256s, round 4), 1 = the role-split loop of rounds 1-3): bit identity on a set of shapes (ragged edges, gate + residual, activation,
two-destination, conv), then TFLOP/s on the headline's shapes with N(0,1) x N(0,0.02) operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def run_forms(M, N, K):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    rpb = M // 2 if M % 2 == 0 else M
    gate = torch.randn(M // rpb, N, device=dev).bfloat16(); resid = torch.randn(M, N, device=dev).bfloat16()
    outs = [ops.gemm(a, w, bias=b), ops.gemm(a, w, bias=b, act=ops.ACT_GELU_TANH, act_n0=(N // 2) // 4 * 4), ops.gemm(a, w, out_f32=True)]
    x = resid.clone()
    ops.gemm(a, w, out=x, bias=b, M=M, lda=K, ldc=N, c_rows_per_batch=rpb, c_batch_stride=rpb * N, gate=gate, resid=x, ldg=N)
    outs.append(x)
    y = torch.zeros((M, N + 4), dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, out=y, bias=b, M=M, lda=K, ldc=N + 4)
    outs.append(y)
    return [o.clone() for o in outs]

ok = True
for (M, N, K) in [(2048, 1024, 256), (2304, 1100, 320), (4100, 3072, 3072), (8192, 768, 1024), (2049, 260, 512), (42696, 3072, 3072)]:
    torch.manual_seed(M + N)
    ops.set_option("gemm_kernel", 2)
    res = {}
    for loop in (1, 0):
        ops.set_option("gemm_t256_loop", loop)
        torch.manual_seed(M + N)
        res[loop] = run_forms(M, N, K)
    same = all(torch.equal(x, y) for x, y in zip(res[0], res[1]))
    fin = all(torch.isfinite(x.float()).all().item() for x in res[0])
    ok &= same and fin
    print(f"bits {M}x{N}x{K}: {'identical' if same else 'DIFFERENT'} finite={fin}", flush=True)
ops.set_option("gemm_kernel", 0)

def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [(32768, 3072, 3072), (32768, 12288, 3072), (42696, 9216, 3072), (42696, 12288, 3072), (42696, 3072, 15360), (42696, 3072, 3072),
          (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    best = {0: 1e9, 1: 1e9}
    for rnd in range(4):
        for loop in (1, 0):
            ops.set_option("gemm_t256_loop", loop)
            best[loop] = min(best[loop], bench(lambda: ops.gemm(A, W, out=C)))
    tf = {k: 2 * M * N * K / v / 1e9 for k, v in best.items()}
    print(f"gemm {M}x{N}x{K}: role-split {tf[1]:.0f}  hand-placed {tf[0]:.0f} TFLOP/s  ({100 * (tf[0] / tf[1] - 1):+.1f} %)", flush=True)
    del A, W, C
ops.set_option("gemm_t256_loop", 0)
print("BITS_OK" if ok else "BITS_FAIL")
