"""Is BASELINE configs[1]'s in-pipeline GEMM rate (B = 1: every launch streams weights no earlier launch of the step touched) a
cold-weight effect?  The same launch over ONE weight matrix (hot in the 256 MB Infinity Cache after the first pass), over a rotation
of NW distinct matrices (> 256 MB together: cold, as in the pipeline), and over the rotation with the NEXT matrix touched (one dword
per 128-B line) by a side stream while the current launch runs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(1536, 3072, 15360), (1536, 21504, 3072), (1536, 3072, 12288), (1536, 9216, 3072), (1536, 12288, 3072), (1536, 3072, 3072)]
side = torch.cuda.Stream()
for (M, N, K) in SHAPES:
    NW = max(2, int(600e6 // (N * K * 2)) + 1)
    A = torch.randn(M, K, device=dev).bfloat16()
    Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NW)]
    bias = torch.randn(N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    touch = [w.view(torch.int32).view(-1, 32)[:, 0] for w in Ws]       # one dword per 128-B line
    def run(mode, iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for i in range(iters):
            w = Ws[0] if mode == "hot" else Ws[i % NW]
            if mode == "prefetch":
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)                       # starts with this launch, not before
                    touch[(i + 1) % NW].sum()
            ops.gemm(A, w, out=C, bias=bias)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters
    res = {}
    for mode in ("hot", "cold", "prefetch"):
        run(mode, NW)
        res[mode] = statistics.median(run(mode, 4 * NW) for _ in range(3))
    fl = 2 * M * N * K / 1e9
    print(f"M={M} N={N} K={K} ({NW} matrices of {N*K*2/1e6:.0f} MB): " + " | ".join(f"{k} {fl/v:.0f} TF/s ({v*1e3:.0f} us)" for k, v in res.items()), flush=True)
