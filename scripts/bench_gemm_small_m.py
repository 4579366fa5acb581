"""GEMM kernels at small M (BASELINE configs[1]: 512 text / 1024 image / 1536 joint rows): t128 (double-buffered 128x128), t256
(persistent 256x256), gemm_bf16_deep<MI, ST> ((32 MI) x 128 tiles, ST-stage ring; option value 10 MI + ST), the policy's pick
(option 0; 1xx = the 192-column tiles of round 3: 100 + 10 MI + ST) and torch.matmul (hipBLASLt) as the yardstick.  Every variant must produce the same bits as t128."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
VARIANTS = [("t128", 1), ("t256", 2)] + [(f"d{c}", c) for c in (42, 43, 32, 33, 22, 23, 24, 13, 14, 113, 123, 133, 143, 132, 142, 134, 144)] + [("policy", 0)]
SHAPES = [(8, 18432, 3072), (8, 9216, 3072), (8, 3072, 256), (64, 3072, 3072), (729, 4096, 1152), (1458, 4304, 1152), (1458, 1152, 4352),
          (512, 3072, 3072), (512, 9216, 3072), (512, 12288, 3072), (512, 3072, 12288),
          (1024, 3072, 3072), (1024, 9216, 3072), (1024, 12288, 3072), (1024, 3072, 12288),
          (1536, 9216, 3072), (1536, 12288, 3072), (1536, 21504, 3072), (1536, 3072, 15360)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
for (M, N, K) in SHAPES:
    NW = max(2, int(600e6 // (N * K * 2)) + 1) if os.environ.get("COLD") else 1      # COLD=1: a rotation of > 600 MB of weight matrices
    A = torch.randn(M, K, device=dev).bfloat16(); Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NW)]
    W = Ws[0]; it = [0]
    def nextw():
        it[0] = (it[0] + 1) % NW
        return Ws[it[0]]
    bias = torch.randn(N, device=dev).bfloat16()
    t = {k: [] for k, _ in VARIANTS}; t["hipblaslt"] = []
    outs = {}
    for rep in range(5):
        for name, code in VARIANTS:
            ops.set_option("gemm_kernel", code)
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            run = lambda: ops.gemm(A, nextw(), out=C, bias=bias)
            if rep == 0:
                it[0] = NW - 1; bench(run, NW if NW > 1 else 3); outs[name] = C.clone()
            t[name].append(bench(run, max(20, 2 * NW)))
        if rep == 0: bench(lambda: torch.matmul(A, W.t()), 3)
        t["hipblaslt"].append(bench(lambda: torch.matmul(A, nextw().t()), max(20, 2 * NW)))
    ops.set_option("gemm_kernel", 0)
    same = all(torch.equal(outs["t128"], o) for o in outs.values())
    fl = 2 * M * N * K / 1e9
    best = min((statistics.median(v), k) for k, v in t.items() if k not in ("hipblaslt", "policy"))
    print(f"M={M} N={N} K={K} bits_equal={same} best={best[1]}: " + " | ".join(f"{k} {fl/statistics.median(v):.0f}" for k, v in t.items()) + " TF/s", flush=True)
