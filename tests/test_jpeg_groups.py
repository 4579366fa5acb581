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
