"""A/B inside one process, interleaved: the top-100 call through the group maxima (two launches, "topk_path" 0) against the
sampled-threshold form (four launches, "topk_path" 1), and the scan-only pass, at the reference's call shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for N in [int(x) for x in os.environ.get("NS", "118287,50000,131072,20000").split(",")]:
    g = torch.Generator(device=dev).manual_seed(0)
    corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
    qs = torch.randn(64, 512, device=dev, generator=g); qs /= qs.norm(dim=-1, keepdim=True)
    if os.environ.get("PLANTED"):        # queries with a near-duplicate in the corpus (bench.py's side field): one key far above the rest
        qs = (corpus[torch.randint(0, N, (64,), generator=g, device=dev)] + 0.02 * torch.randn(64, 512, generator=g, device=dev)).contiguous()
    for Q in (1, 4, 16, 32, 64):
        q = qs[:Q].contiguous()
        t = {0: [], 1: []}
        for rep in range(3):
            for path in (0, 1):
                ops.set_option("topk_path", path)
                t[path].append(bench(lambda: ops.cosine_topk(corpus, q, 100)))
        ops.set_option("topk_path", 0)
        D0, I0 = ops.cosine_topk(corpus, q, 100)
        ops.set_option("topk_path", 1)
        D1, I1 = ops.cosine_topk(corpus, q, 100)
        ops.set_option("topk_path", 0)
        sc = ops.cosine_scores(corpus, q)
        scan = bench(lambda: ops.cosine_scores(corpus, q, out=sc))
        print(f"N={N} Q={Q}: group-maxima path {min(t[0]):.1f} us ({', '.join(f'{x:.1f}' for x in t[0])}) | sampled-threshold path {min(t[1]):.1f} us "
              f"({', '.join(f'{x:.1f}' for x in t[1])}) | scan only {scan:.1f} us | same (D, I): {bool(torch.equal(D0, D1) and torch.equal(I0, I1))}", flush=True)
