"""20 launches of ONE shape of the retrieval scan (and of the whole top-k call) for a rocprofv3 --pmc pass:
    python scripts/pmc_scan_shape.py N Q [scan|topk]
A pass over one shape only, so the per-kernel counter sums divide by launches of that shape (round 2's summary averaged
N = 1 000 / 118 287 / 1 000 000 launches of bench_topk.py into one figure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
N, Q = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "scan"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
corpus = torch.randn(N, 512, device=dev, generator=g)
qs = torch.randn(Q, 512, device=dev, generator=g)
sc = ops.cosine_scores(corpus, qs) if what == "scan" else None
for _ in range(20):
    if what == "scan":
        ops.cosine_scores(corpus, qs, out=sc)
    else:
        ops.cosine_topk(corpus, qs, 100)
torch.cuda.synchronize()
