"""drag_gemm_bf16_pair on the Linear pairs of a Flux double block at batch 1 (BASELINE configs[1]: 1024 image + 512 text rows):
two launches ("gemm_pair" 1) against one merged launch (2) under the policy and under every forced kernel, TFLOP/s over both problems.
PAIRS=M1xM2xNxK,... overrides the list.  COLD=1: every launch reads weights no recent launch touched (a rotation of > 600 MB of
matrices, as inside a batch-1 forward: the 256 MB Infinity Cache holds none of them); EPI=1: gate + residual epilogue (in place)."""
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
PAIRS = [(1024, 512, 9216, 3072), (1024, 512, 3072, 3072), (1024, 512, 12288, 3072), (1024, 512, 3072, 12288),
         (4096, 1024, 9216, 3072), (4096, 1024, 3072, 12288), (32768 + 9928, 4096, 9216, 3072)]
if os.environ.get("PAIRS"):
    PAIRS = [tuple(int(v) for v in s.split("x")) for s in os.environ["PAIRS"].split(",")]
CODES = [0, 1, 2, 23, 32, 33, 42, 43, 113, 123, 133, 143, 132, 142, 134, 144]
for (M1, M2, N, K) in PAIRS:
    COLD, EPI = bool(os.environ.get("COLD")), bool(os.environ.get("EPI"))
    NW = max(2, int(600e6 // (2 * N * K * 2)) + 1) if COLD else 1
    A1, A2 = torch.randn(M1, K, device=dev).bfloat16(), torch.randn(M2, K, device=dev).bfloat16()
    W1 = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NW)]
    W2 = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NW)]
    b1, b2 = torch.randn(N, device=dev).bfloat16(), torch.randn(N, device=dev).bfloat16()
    C1, C2 = torch.empty(M1, N, device=dev, dtype=torch.bfloat16), torch.empty(M2, N, device=dev, dtype=torch.bfloat16)
    g1, g2 = torch.randn(1, N, device=dev).bfloat16(), torch.randn(1, N, device=dev).bfloat16()
    R1, R2 = torch.randn(M1, N, device=dev).bfloat16(), torch.randn(M2, N, device=dev).bfloat16()
    it = [0]
    def run():
        i = it[0] = (it[0] + 1) % NW
        if EPI:
            ops.gemm_pair(dict(a=A1, w=W1[i], out=C1, bias=b1, gate=g1, resid=R1, ldg=N), dict(a=A2, w=W2[i], out=C2, bias=b2, gate=g2, resid=R2, ldg=N))
        else:
            ops.gemm_pair(dict(a=A1, w=W1[i], out=C1, bias=b1), dict(a=A2, w=W2[i], out=C2, bias=b2))
    names = ["two"] + [f"k{c}" for c in CODES if not (c > 100 and N % 192)]
    t = {n: [] for n in names}
    ref = None
    same = True
    for rep in range(5):
        for n in names:
            ops.set_option("gemm_pair", 1 if n == "two" else 2)
            ops.set_option("gemm_kernel", 0 if n == "two" else int(n[1:]))
            if rep == 0:
                C1.zero_(); C2.zero_(); it[0] = NW - 1; bench(run, NW)
                if ref is None: ref = (C1.clone(), C2.clone())
                else: same = same and torch.equal(ref[0], C1) and torch.equal(ref[1], C2)
            t[n].append(bench(run, max(20, 2 * NW)))
    ops.set_option("gemm_kernel", 0); ops.set_option("gemm_pair", 0)
    fl = 2 * (M1 + M2) * N * K / 1e9
    best = min((statistics.median(v), n) for n, v in t.items() if n not in ("two", "k0"))
    print(f"M={M1}+{M2} N={N} K={K} bits_equal={same} best={best[1]}: " + " | ".join(f"{n} {fl/statistics.median(v):.0f}" for n, v in t.items()) + " TF/s", flush=True)
