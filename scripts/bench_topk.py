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
for N in (1000, 118287, 1000000):
    corpus = torch.randn(N, 512, device=dev); qs = torch.randn(64, 512, device=dev)
    for Q in (1, 16, 64):
        ms = bench(lambda: ops.cosine_topk(corpus, qs[:Q], 100))
        bytes_ = N * 512 * 4 * ((Q + 15) // 16) + Q * 512 * 4 + Q * 100 * 12
        print(f"topk N={N} Q={Q}: {ms*1e3:.1f} us/call  {bytes_/ms/1e6:.0f} GB/s algorithmic (scan passes x corpus bytes)", flush=True)
