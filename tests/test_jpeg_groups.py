"""DecodedBatch.groups() with a rejected file INSIDE a size class (status 10 is set after the output slots are planned,
so the file keeps its slot): every row of every group must still be image(idx[r]).  Host-only: the batch object is plain
bookkeeping over a byte buffer (one truncated JPEG used to shift every later same-size image's pixels)."""
import numpy as np
import torch

from domain_rag_amd import jpeg


def _batch(sizes, rejected):
    n = len(sizes)
    info = np.zeros((n, jpeg.INFO_WORDS), dtype=np.int32)
    for i, (h, w) in enumerate(sizes):
        info[i, 1], info[i, 2] = w, h
    order = np.lexsort((np.arange(n), info[:, 1], info[:, 2])).astype(np.int64)
    off = np.zeros(n, dtype=np.int64)
    o = 0
    for i in order:
        off[i] = o
        o += sizes[i][0] * sizes[i][1] * 3
    out = torch.empty(o, dtype=torch.uint8)
    for i in range(n):                                  # image i is filled with the byte i + 1
        out[off[i]: off[i] + sizes[i][0] * sizes[i][1] * 3] = i + 1
    for i in rejected:
        info[i, 0] = 10
    return jpeg.DecodedBatch(info, out, off, order)


def _check(batch, sizes, rejected):
    seen = []
    for (h, w), idx, imgs in batch.groups():
        assert imgs.shape == (len(idx), h, w, 3)
        for r, i in enumerate(idx.tolist()):
            assert sizes[i] == (h, w)
            assert torch.equal(imgs[r], batch.image(i)) and int(imgs[r].min()) == int(imgs[r].max()) == i + 1, (i, r)
        seen += idx.tolist()
    assert sorted(seen) == [i for i in range(len(sizes)) if i not in rejected]


def test_rejected_file_in_the_middle_of_a_size_class():
    sizes = [(4, 6)] * 3
    b = _batch(sizes, rejected=[1])
    _check(b, sizes, [1])
    assert len(list(b.groups())) == 2                   # the class splits around the dead slot


def test_rejected_files_anywhere():
    rng = np.random.default_rng(0)
    for trial in range(50):
        n = int(rng.integers(1, 24))
        sizes = [[(4, 6), (6, 4), (2, 2), (4, 6)][int(rng.integers(4))] for _ in range(n)]
        rejected = [i for i in range(n) if rng.random() < 0.3]
        _check(_batch(sizes, rejected), sizes, rejected)


def test_no_rejection_keeps_one_group_per_size():
    sizes = [(4, 6), (2, 2), (4, 6), (4, 6), (2, 2)]
    b = _batch(sizes, [])
    _check(b, sizes, [])
    assert len(list(b.groups())) == 2


def test_budget_bounds_cut_a_chunk_by_pixels():
    """consecutive ranges whose pixel sums stay within the budget; a file above the budget gets a range of its own; undecodable
    files (0 pixels) ride along; every index is covered exactly once, in order"""
    px = np.array([5, 5, 5, 20, 1, 1, 0, 0, 9], dtype=np.int64)
    assert jpeg.budget_bounds(px, 10) == [(0, 2), (2, 3), (3, 4), (4, 8), (8, 9)]
    assert jpeg.budget_bounds(px, 1000) == [(0, 9)]
    assert jpeg.budget_bounds(np.zeros(4, dtype=np.int64), 1) == [(0, 4)]
    assert jpeg.budget_bounds(np.array([], dtype=np.int64), 1) == []
    rng = np.random.default_rng(0)
    for _ in range(50):
        p = rng.integers(0, 50, int(rng.integers(1, 40)))
        b = jpeg.budget_bounds(p, 60)
        assert b[0][0] == 0 and b[-1][1] == len(p) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all(p[a:e].sum() <= 60 or e - a == 1 for a, e in b)


def test_staged_files_slice_is_a_batch_of_its_own():
    buf = torch.arange(100, dtype=torch.uint8)
    st = jpeg.StagedFiles(buf, np.array([0, 10, 10, 35, 60], dtype=np.int64), {1: OSError("unreadable")})
    sub = st.slice(1, 3)
    assert len(sub) == 2 and sub.offsets.tolist() == [0, 0, 25] and list(sub.errors) == [0]
    assert sub.file_bytes(1) == bytes(range(10, 35))
    assert st.slice(3, 4).file_bytes(0) == bytes(range(35, 60)) and st.slice(3, 4).errors == {}
