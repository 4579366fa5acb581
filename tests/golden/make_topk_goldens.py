"""Generate tests/golden/topk_golden.npz with the C oracle (oracle/topk.c): seeded unit-norm corpus + queries with
planted exact ties.  Small fixture (inputs are regenerated from the seed; only expected outputs are stored)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import retrieval as oret  # noqa: E402


def inputs():
    rng = np.random.default_rng(20260928)
    corpus = rng.standard_normal((1000, 512)).astype(np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    corpus[500] = corpus[17]; corpus[999] = corpus[17]; corpus[3] = corpus[640]      # exact ties
    q = corpus[[17, 640, 5, 77]] * 0.9 + 0.1 * rng.standard_normal((4, 512)).astype(np.float32)
    q = np.concatenate([q, rng.standard_normal((12, 512)).astype(np.float32)], 0)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return corpus, q.astype(np.float32)


if __name__ == "__main__":
    c, q = inputs()
    D, I = oret.cosine_topk(c, q, 100)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "topk_golden.npz")
    np.savez_compressed(out, D=D, I=I.astype(np.int32))
    print("wrote", out, os.path.getsize(out))
