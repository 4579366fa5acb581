"""one shape of the top-k call under rocprofv3 --kernel-trace: python scripts/profile_topk.py N Q  (per-kernel split of the call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
N, Q = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
qs = torch.randn(Q, 512, device=dev, generator=g); qs /= qs.norm(dim=-1, keepdim=True)
for _ in range(30):
    ops.cosine_topk(corpus, qs, 100)
torch.cuda.synchronize()
