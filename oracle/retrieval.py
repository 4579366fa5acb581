"""oracle/retrieval.py — Python face of oracle/topk.c + numpy restatements of the retrieval host math.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "liboracle_topk.so")
_lib = None


def build() -> str:
    src = os.path.join(HERE, "topk.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_cosine_topk.restype = ctypes.c_int
        _lib.oracle_cosine_topk.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_ip_scores.restype = None
        _lib.oracle_ip_scores.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p]
    return _lib


def cosine_topk(corpus: np.ndarray, queries: np.ndarray, k: int):
    """(D float32 [Q,k] desc, I int64 [Q,k]) — IndexFlatIP.search semantics in the build's defined order
    (retrieval/clip100_resnet_style_all_shots.py:425-434)."""
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    N, d = corpus.shape
    Q = queries.shape[0]
    D = np.empty((Q, k), dtype=np.float32)
    I = np.empty((Q, k), dtype=np.int64)
    rc = lib().oracle_cosine_topk(corpus.ctypes.data, queries.ctypes.data, N, d, Q, k, D.ctypes.data, I.ctypes.data)
    if rc != 0:
        raise ValueError(f"oracle_cosine_topk rc={rc}")
    return D, I


def ip_scores(corpus: np.ndarray, queries: np.ndarray) -> np.ndarray:
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    out = np.empty((queries.shape[0], corpus.shape[0]), dtype=np.float32)
    lib().oracle_ip_scores(corpus.ctypes.data, queries.ctypes.data, corpus.shape[0], corpus.shape[1],
                           queries.shape[0], out.ctypes.data)
    return out
