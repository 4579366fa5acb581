"""The C oracle for the retrieval order (oracle/topk.c): agrees with an independent float64 ranking, honours the
tie rule, reproduces the committed golden vectors (config 1 of BASELINE: 1k corpus, top-100, 16 queries)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_oracle_matches_golden_and_float64():
    from make_topk_goldens import inputs
    from oracle import retrieval as oret
    corpus, q = inputs()
    D, I = oret.cosine_topk(corpus, q, 100)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "topk_golden.npz"))
    assert np.array_equal(I, gold["I"].astype(np.int64)) and np.array_equal(D, gold["D"])
    # independent check: float64 scores, stable argsort (ties -> lower index); fp32 rounding can only reorder
    # candidates whose float64 scores are within ~1e-6
    s64 = q.astype(np.float64) @ corpus.astype(np.float64).T
    order = np.argsort(-s64, axis=1, kind="stable")[:, :100]
    assert np.abs(np.take_along_axis(s64, I, 1) - np.take_along_axis(s64, order, 1)).max() < 2e-6
    assert (D[:, :-1] >= D[:, 1:]).all()
    # planted exact duplicates: 17 == 500 == 999 and 3 == 640 -> equal scores, ascending index
    for row in range(q.shape[0]):
        pos = {int(i): p for p, i in enumerate(I[row])}
        if 17 in pos and 500 in pos and 999 in pos:
            assert pos[17] + 1 == pos[500] and pos[500] + 1 == pos[999] and D[row, pos[17]] == D[row, pos[999]]
        if 3 in pos and 640 in pos:
            assert pos[3] + 1 == pos[640]
    assert I[0, 0] == 17 and I[1, 0] == 3


def test_oracle_score_order_is_the_documented_fma_chain():
    from oracle import retrieval as oret
    rng = np.random.default_rng(1)
    c = rng.standard_normal((5, 64)).astype(np.float32)
    q = rng.standard_normal((2, 64)).astype(np.float32)
    s = oret.ip_scores(c, q)
    import math
    for n in range(5):
        for j in range(2):
            acc = np.float32(0)
            for blk in range(4):
                for ss in range(4):
                    for g in range(4):
                        k = 16 * blk + 4 * g + ss
                        # fma in double is exact for the product; one rounding to fp32 per step
                        acc = np.float32(float(c[n, k]) * float(q[j, k]) + float(acc))
            assert abs(float(acc) - float(s[j, n])) <= 1e-6 * max(1.0, abs(float(acc)))


def test_oracle_padding_and_small_n():
    from oracle import retrieval as oret
    c = np.eye(4, 16, dtype=np.float32)
    D, I = oret.cosine_topk(c, c[:1], 6)
    assert I[0].tolist() == [0, 1, 2, 3, -1, -1] and D[0, 0] == 1.0 and D[0, 4] == -np.finfo(np.float32).max
