"""A/B inside one process, interleaved: the d = 512 scan with the query tile in registers (ip_scan_q512_kernel, "topk_qreg" 0) against the
scan that reads it from LDS for every corpus chunk (ip_scan_kernel, "topk_qreg" 1) — the scan alone and the whole top-100 call, at the
reference's corpus (N = 118 287) and past the Infinity Cache (N = 1 000 000).  Same (D, I) and same score bits are asserted."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for N in [int(x) for x in os.environ.get("NS", "118287,1000000").split(",")]:
    g = torch.Generator(device=dev).manual_seed(0)
    corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
    qs = (corpus[torch.randint(0, N, (64,), generator=g, device=dev)] + 0.02 * torch.randn(64, 512, generator=g, device=dev)).contiguous()
    iters = 50 if N < 500000 else 12
    for Q in (1, 16, 32, 64):
        q = qs[:Q].contiguous()
        call, scan = {0: [], 1: []}, {0: [], 1: []}
        sc = ops.cosine_scores(corpus, q)
        ref = {}
        for v in (0, 1):
            ops.set_option("topk_qreg", v)
            ops.cosine_scores(corpus, q, out=sc)
            D, I = ops.cosine_topk(corpus, q, 100)
            ref[v] = (sc[:, :N].clone(), D, I)          # (columns past N: the ragged last group and the row padding, written by nobody)
        same = torch.equal(ref[0][0], ref[1][0]) and torch.equal(ref[0][1], ref[1][1]) and torch.equal(ref[0][2], ref[1][2])
        for rep in range(3):
            for v in (0, 1):
                ops.set_option("topk_qreg", v)
                call[v].append(bench(lambda: ops.cosine_topk(corpus, q, 100), iters))
                scan[v].append(bench(lambda: ops.cosine_scores(corpus, q, out=sc), iters))
        ops.set_option("topk_qreg", 0)
        gb = (N * 512 * 4 + Q * 512 * 4) / 1e3
        print(f"N={N} Q={Q}: registers: call {min(call[0]):.1f} us scan {min(scan[0]):.1f} us ({gb / min(scan[0]) / 1e3:.2f} TB/s) | LDS image: call {min(call[1]):.1f} us "
              f"scan {min(scan[1]):.1f} us ({gb / min(scan[1]) / 1e3:.2f} TB/s) | same scores and (D, I): {same}", flush=True)
