"""one top-100 call shape in a loop (for rocprofv3 --kernel-trace --stats): Q and N from the environment"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
N, Q = int(os.environ.get("N", "118287")), int(os.environ.get("Q", "1"))
ops.set_option("topk_path", int(os.environ.get("PATH_", "0")))
g = torch.Generator(device=dev).manual_seed(0)
corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
q = torch.randn(Q, 512, device=dev, generator=g); q /= q.norm(dim=-1, keepdim=True)
for _ in range(200): ops.cosine_topk(corpus, q, 100)
torch.cuda.synchronize()
