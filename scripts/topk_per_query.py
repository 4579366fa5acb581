"""the top-100 call one query at a time (N = 118 287): a data-dependent slow path of the selection shows as a slow query"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
N = 118287
g = torch.Generator(device=dev).manual_seed(0)
corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
qs = torch.randn(64, 512, device=dev, generator=g); qs /= qs.norm(dim=-1, keepdim=True)
sc = ops.cosine_scores(corpus, qs[:32].contiguous())[:, :N]
neg = (sc[:, :N // 16 * 16].reshape(32, -1, 16).max(-1).values < 0).sum(-1).tolist()
ts = [bench(lambda: ops.cosine_topk(corpus, qs[i:i + 1].contiguous(), 100)) for i in range(32)]
print("call us per query (groups of 16 rows whose scores are all negative): " + " ".join(f"{i}:{t:.1f}({n})" for i, (t, n) in enumerate(zip(ts, neg))))
for Q in (1, 8, 16, 32, 64):
    q = qs[:Q].contiguous()
    print(f"Q={Q}: call {bench(lambda: ops.cosine_topk(corpus, q, 100)):.1f} us")
