"""random (N, d, Q, k) with duplicated rows (exact ties), planted NaN / inf rows and random measurement options, against oracle/topk.c:
scores and indices must be the oracle's bit for bit.   python scripts/fuzz_topk.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from domain_rag_amd import ops
from oracle import retrieval as oret
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for c in range(cases):
    d = int(rng.choice([64, 128, 256, 512]))
    N = int(rng.choice([rng.integers(1, 600), rng.integers(600, 9000), rng.integers(8193, 70000)]))
    Q = int(rng.choice([1, rng.integers(1, 17), rng.integers(17, 65), rng.integers(65, 140)]))
    k = int(rng.choice([1, rng.integers(1, 129), rng.integers(129, 2049)]))
    pool = rng.standard_normal((max(1, int(N * rng.choice([1.0, 0.3, 0.02]))), d)).astype(np.float32)
    corpus = pool[rng.integers(0, len(pool), N)]                      # duplicates -> exact ties, resolved by index
    if rng.random() < 0.3 and N > 10:
        corpus[rng.integers(0, N)] = np.nan; corpus[rng.integers(0, N)] = np.inf; corpus[rng.integers(0, N), 0] = -np.inf
    q = rng.standard_normal((Q, d)).astype(np.float32)
    opts = {"topk_qt": int(rng.choice([0, 2, 4])), "topk_grid": int(rng.choice([0, 256, 1024, 2048])), "topk_depth": int(rng.choice([0, 3])),
            "topk_dense_sample": int(rng.integers(0, 2)), "topk_select": int(rng.choice([0, 256, 1024]))}
    for n_, v in opts.items():
        ops.set_option(n_, v)
    D, I = ops.cosine_topk(torch.from_numpy(corpus).to(dev), torch.from_numpy(q).to(dev), k)
    for n_ in opts:
        ops.set_option(n_, 0)
    Dr, Ir = oret.cosine_topk(corpus, q, k)
    ok = np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy().view(np.uint32), Dr.view(np.uint32))
    bad += int(not ok)
    print(f"case {c}: N={N} d={d} Q={Q} k={k} pool={len(pool)} {opts} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print("FUZZ", "OK" if bad == 0 else f"FAILED ({bad})")
