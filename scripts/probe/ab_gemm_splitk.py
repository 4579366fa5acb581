"""round 5: split-K ("gemm_splitk" 0 = policy, 1 = never) on the batch-1 Linears of few tiles and long K, and BASELINE configs[1] with and without it"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def tm(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in [(1536, 3072, 15360), (1024, 3072, 12288), (512, 3072, 12288), (1024, 3072, 3072), (2048, 3072, 15360), (1536, 3072, 8192)]:
    g = torch.Generator(device=dev).manual_seed(K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(N, device=dev, generator=g).bfloat16(); gate = torch.randn(1, N, device=dev, generator=g).bfloat16(); resid = torch.randn(M, N, device=dev, generator=g).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=bias, gate=gate, resid=resid, ldg=N, c_rows_per_batch=M, c_batch_stride=M * N)
    t = {}
    for v in (1, 0, 2, 3, 4, 6, 8):
        ops.set_option("gemm_splitk", v)
        try: t[v] = statistics.median(tm(lambda: ops.gemm(A, W, out=C, **kw)) for _ in range(3))
        except Exception as ex: t[v] = float("nan")
    ops.set_option("gemm_splitk", 0)
    fl = 2 * M * N * K / 1e6
    print(f"M={M} N={N} K={K}: never {t[1]:.1f} us ({fl / t[1]:.0f} TF/s) | policy {t[0]:.1f} us ({fl / t[0]:.0f}) | forced slices: " + " ".join(f"{v}: {t[v]:.1f}" for v in (2, 3, 4, 6, 8)), flush=True)
res = {0: [], 1: []}
for rep in range(3):
    for v in (1, 0):
        ops.set_option("gemm_splitk", v)
        res[v].append(bench.side_config1(dev)["ms_per_image"])
ops.set_option("gemm_splitk", 0)
print(f"configs[1]: never {min(res[1]):.2f} ms per image | policy {min(res[0]):.2f} ms ({100 * (min(res[1]) / min(res[0]) - 1):+.1f} %)")
# the double blocks' ff down-projections arrive as a PAIR (image 1024 rows + text 512 rows): merged launch against two split ones
M1, M2, N, K = 1024, 512, 3072, 12288
g = torch.Generator(device=dev).manual_seed(7)
mk = lambda M: dict(a=torch.randn(M, K, device=dev, generator=g).bfloat16(), w=(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16(),
                    out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), bias=torch.randn(N, device=dev, generator=g).bfloat16(),
                    gate=torch.randn(1, N, device=dev, generator=g).bfloat16(), resid=torch.randn(M, N, device=dev, generator=g).bfloat16(), ldg=N,
                    c_rows_per_batch=M, c_batch_stride=M * N)
f, s2 = mk(M1), mk(M2)
for v in (1, 0):
    ops.set_option("gemm_splitk", v)
    tp = statistics.median(tm(lambda: ops.gemm_pair(f, s2)) for _ in range(3))
    ts = statistics.median(tm(lambda: (ops.gemm(**f), ops.gemm(**s2))) for _ in range(3))
    print(f"pair (1024 + 512) x 3072 x 12288, gemm_splitk = {v}: drag_gemm_bf16_pair {tp:.1f} us | two drag_gemm_bf16 {ts:.1f} us")
ops.set_option("gemm_splitk", 0)
