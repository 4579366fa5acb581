"""round 5: the qreg A/B printed "same scores and (D, I): False" at N = 118 287, Q = 32 / 64 — which output differs, from which kernel, and is it stable?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from domain_rag_amd import ops
from oracle import retrieval as oret
dev = torch.device("cuda:0")
N = 118287
g = torch.Generator(device=dev).manual_seed(0)
corpus = torch.randn(N, 512, device=dev, generator=g); corpus /= corpus.norm(dim=-1, keepdim=True)
qs = (corpus[torch.randint(0, N, (64,), generator=g, device=dev)] + 0.02 * torch.randn(64, 512, generator=g, device=dev)).contiguous()
for Q in (16, 32, 64):
    q = qs[:Q].contiguous()
    Dr, Ir = oret.cosine_topk(corpus.cpu().numpy(), q.cpu().numpy(), 100)
    out = {}
    for v in (0, 1):
        ops.set_option("topk_qreg", v)
        runs = []
        for rep in range(4):
            sc = torch.zeros(Q, (N + 63) // 64 * 64, device=dev)
            ops.cosine_scores(corpus, q, out=sc)
            D, I = ops.cosine_topk(corpus, q, 100)
            runs.append((sc[:, :N].clone(), D.clone(), I.clone()))
        stable = all(torch.equal(runs[0][i], r[i]) for r in runs[1:] for i in range(3))
        ok = np.array_equal(runs[0][1].cpu().numpy(), Dr) and np.array_equal(runs[0][2].cpu().numpy(), Ir)
        out[v] = runs[0]
        print(f"Q={Q} topk_qreg={v}: stable over 4 runs {stable}; (D, I) == oracle {ok}", flush=True)
    ds = (out[0][0] != out[1][0])
    print(f"   scores differ between the kernels at {int(ds.sum())} of {ds.numel()} places; D differ {int((out[0][1] != out[1][1]).sum())}; I differ {int((out[0][2] != out[1][2]).sum())}")
    if ds.any():
        idx = ds.nonzero()[:8]
        print("   first places (query, row):", idx.tolist())
ops.set_option("topk_qreg", 0)
