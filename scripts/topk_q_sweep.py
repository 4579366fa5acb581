import sys; sys.path.insert(0, '.')
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
out = []
for Q in (1, 2, 4, 6, 8, 9, 10, 12, 16, 17, 24, 32, 48, 64):
    q = qs[:Q].contiguous()
    sc = ops.cosine_scores(corpus, q)
    out.append(f"Q={Q}: call {bench(lambda: ops.cosine_topk(corpus, q, 100)):.1f} us, scan only {bench(lambda: ops.cosine_scores(corpus, q, out=sc)):.1f} us")
print("\n".join(out))
