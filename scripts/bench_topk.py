"""top-k call and scan-only pass, per Q and workgroup cap (drag_set_option("topk_grid")); algorithmic bytes = SURVEY 8(d):
N*512*4 + Q*512*4 + Q*k*12 for the call (ONE corpus pass for Q <= 64), + the score rows written for the scan-only entry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
grids = [int(x) for x in os.environ.get("GRIDS", "0").split(",")]
for N in [int(x) for x in os.environ.get("NS", "1000,118287,1000000").split(",")]:
    g = torch.Generator(device=dev).manual_seed(0)
    corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
    qs = torch.randn(128, 512, device=dev, generator=g); qs /= qs.norm(dim=-1, keepdim=True)
    for grid, depth in [(g_, d_) for g_ in grids for d_ in [int(x) for x in os.environ.get("DEPTHS", "0").split(",")]]:
        ops.set_option("topk_grid", grid); ops.set_option("topk_depth", depth)
        ops.set_option("topk_select", int(os.environ.get("SELECT", "0"))); ops.set_option("topk_qt", int(os.environ.get("QT", "0")))
        for Q in (1, 16, 32, 64, 128):
            q = qs[:Q].contiguous()
            ms = bench(lambda: ops.cosine_topk(corpus, q, 100))
            passes = (Q + 63) // 64
            b = N * 512 * 4 * passes + Q * 512 * 4 + Q * 100 * 12
            line = f"N={N} grid={grid} depth={depth} Q={Q}: top-100 call {ms*1e3:.1f} us = {b/ms/1e6:.0f} GB/s"
            if Q <= 64:
                sc = ops.cosine_scores(corpus, q)
                ms2 = bench(lambda: ops.cosine_scores(corpus, q, out=sc))
                b2 = N * 512 * 4 + Q * 512 * 4 + Q * sc.shape[1] * 4
                line += f" | scan only {ms2*1e3:.1f} us = {b2/ms2/1e6:.0f} GB/s, {2*N*512*Q/ms2/1e9:.1f} TFLOP/s f32"
            print(line, flush=True)
ops.set_option("topk_grid", 0); ops.set_option("topk_depth", 0); ops.set_option("topk_select", 0); ops.set_option("topk_qt", 0)
