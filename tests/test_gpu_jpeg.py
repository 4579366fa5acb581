"""GPU JPEG decode (csrc/jpeg.hip) against PIL itself: Image.open(f).convert("RGB") byte for byte — the oracle here is the real
dependency the reference decodes with (retrieval/clip100_resnet_style_all_shots.py:270-281), not a restatement."""
import io

import numpy as np
import pytest
import torch
from PIL import Image, ImageFile

pytestmark = pytest.mark.gpu


def natural_image(rng, h, w):
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16) + rng.integers(-20, 20, (h, w, 3))
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


def encode(im, **kw):
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, im.size[0] * im.size[1] * 4)
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    return bio.getvalue()


def pil_rgb(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_mixed_batch_is_byte_identical_to_pil(gpu):
    """one batch mixing sizes, subsamplings, qualities, optimised tables, grey images, restart markers and files the device
    path must hand back (arithmetic-coded, CMYK, a PNG, a truncated header); progressive files decode on the device (round 3)"""
    from domain_rag_amd import jpeg
    rng = np.random.default_rng(0)
    blobs, want = [], []
    for (w, h) in [(640, 480), (480, 640), (33, 17), (101, 77), (8, 8), (17, 40), (500, 375), (5, 3), (16, 1), (640, 480)]:
        for sub in (0, 1, 2):
            for q, kw in ((30, {}), (75, {"optimize": True}), (95, {}), (100, {"optimize": True})):
                blobs.append(encode(natural_image(rng, h, w), quality=q, subsampling=sub, **kw))
        blobs.append(encode(natural_image(rng, h, w).convert("L"), quality=80))
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
        blobs.append(encode(natural_image(rng, 90, 130), quality=80, subsampling=2, **kw))
    blobs.append(encode(Image.fromarray(rng.integers(0, 256, (96, 120, 3), dtype=np.uint8)), quality=5))        # noise: saturation
    for (w, h) in [(640, 480), (33, 17), (101, 77), (8, 8), (500, 375)]:       # SOF2: libjpeg's default progression, every scan type
        for sub, kw in ((0, {}), (1, {"optimize": True}), (2, {}), (2, {"restart_marker_blocks": 5})):
            blobs.append(encode(natural_image(rng, h, w), quality=int(rng.integers(30, 98)), subsampling=sub, progressive=True, **kw))
        blobs.append(encode(natural_image(rng, h, w).convert("L"), quality=80, progressive=True))
    n_ok = len(blobs)
    arith = bytearray(encode(natural_image(rng, 40, 56), quality=80, progressive=True))
    arith[arith.index(b"\xff\xc2") + 1] = 0xCA                                  # SOF10: arithmetic coding
    rejected = [(bytes(arith), 3),
                (encode(natural_image(rng, 40, 56).convert("CMYK"), quality=80), 5),
                (b"\x89PNG\r\n\x1a\n" + bytes(40), 1), (blobs[0][:40], 2)]
    blobs += [b for b, _ in rejected]
    batch = jpeg.decode_files(blobs, gpu)
    assert len(batch) == len(blobs)
    assert (batch.status[:n_ok] == 0).all(), [(i, int(s)) for i, s in enumerate(batch.status[:n_ok]) if s]
    assert batch.status[n_ok:].tolist() == [s for _, s in rejected]
    for i in range(n_ok):
        ref = pil_rgb(blobs[i])
        got = batch.image(i).cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), (i, ref.shape, int(np.abs(got.astype(int) - ref).max()))
    assert all(batch.image(i) is None for i in range(n_ok, len(blobs)))
    # groups(): every decodable image exactly once, same-size images as one dense slab with the same bytes
    seen = []
    for (h, w), idx, imgs in batch.groups():
        assert imgs.shape == (len(idx), h, w, 3) and imgs.is_contiguous()
        for k, i in enumerate(idx.tolist()):
            assert torch.equal(imgs[k], batch.image(i))
        seen += idx.tolist()
    assert sorted(seen) == list(range(n_ok))


def test_decode_is_independent_of_the_batch(gpu):
    """an image decodes to the same bytes alone, first, last or in the middle of a batch (per-lane state only)"""
    from domain_rag_amd import jpeg
    rng = np.random.default_rng(1)
    files = [encode(natural_image(rng, 120, 160), quality=int(rng.integers(20, 98)), subsampling=int(rng.integers(0, 3))) for _ in range(70)]
    full = jpeg.decode_files(files, gpu)
    alone = jpeg.decode_files([files[37]], gpu)
    assert torch.equal(alone.image(0), full.image(37))
    rev = jpeg.decode_files(files[::-1], gpu)
    for i in (0, 1, 63, 64, 69):                      # lanes of both waves, both ends
        assert torch.equal(rev.image(len(files) - 1 - i), full.image(i))


def test_decoded_pixels_feed_the_clip_preprocess_unchanged(gpu):
    """decode -> PIL-exact resize + centre crop on the device == PIL decode -> PIL resize -> crop, byte for byte: the
    embedding (and the top-k) cannot tell which route produced the crop"""
    from domain_rag_amd import jpeg, resample
    rng = np.random.default_rng(2)
    files = [encode(natural_image(rng, 480, 640), quality=90, subsampling=2) for _ in range(3)] + \
            [encode(natural_image(rng, 333, 500), quality=85, subsampling=1)]
    batch = jpeg.decode_files(files, gpu)
    crops = torch.empty((len(files), 224, 224, 3), dtype=torch.uint8, device=gpu)
    for (h, w), idx, imgs in batch.groups():
        out = resample.clip_preprocess_u8(imgs)
        crops[torch.from_numpy(idx).to(gpu)] = out
    for i, f in enumerate(files):
        im = Image.open(io.BytesIO(f)).convert("RGB")
        nw, nh, box = resample.clip_resize_plan(*im.size)
        ref = np.asarray(im.resize((nw, nh), Image.BICUBIC).crop(box))
        assert np.array_equal(crops[i].cpu().numpy(), ref), i


def test_argument_errors(gpu):
    from domain_rag_amd import jpeg
    with pytest.raises(ValueError):
        jpeg.decode_files([], gpu)
    with pytest.raises(RuntimeError):
        jpeg.decode_files([b"x"], "cpu")
    b = jpeg.decode_files([b"not a jpeg at all"], gpu)
    assert b.status.tolist() == [1] and b.image(0) is None and list(b.groups()) == []


def test_damaged_files_never_fault_the_gpu(gpu):
    """1 500 mutated files in one batch (byte flips, damaged headers, truncations, stray markers): every one gets a status, the
    kernels stay inside their buffers (a fault would take the process down), and what is reported decodable has its header's
    size.  The same mutations run under AddressSanitizer on the host build (tests/test_jpeg_core_host.py)."""
    from domain_rag_amd import jpeg
    rng = np.random.default_rng(11)
    seeds = [encode(natural_image(rng, h, w), quality=int(rng.integers(20, 98)), subsampling=sub, **kw)
             for (w, h) in ((64, 48), (33, 17), (120, 90)) for sub in (0, 1, 2) for kw in ({}, {"optimize": True}, {"restart_marker_blocks": 2})]
    seeds += [encode(natural_image(rng, h, w), quality=int(rng.integers(20, 98)), subsampling=sub, progressive=True, **kw)
              for (w, h) in ((64, 48), (33, 17)) for sub in (0, 2) for kw in ({}, {"restart_marker_blocks": 3})]     # the SOF2 scan walker too
    files = []
    for i in range(1500):
        f = bytearray(seeds[int(rng.integers(len(seeds)))])
        kind = i % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 9))):
                f[int(rng.integers(len(f)))] = int(rng.integers(256))
        elif kind == 1:
            for _ in range(int(rng.integers(1, 7))):
                f[int(rng.integers(min(len(f), 700)))] = int(rng.integers(256))
        elif kind == 2:
            f = f[: int(rng.integers(1, len(f)))]
        elif kind == 3:
            for _ in range(int(rng.integers(1, 7))):
                p = len(f) // 2 + int(rng.integers(len(f) // 2))
                f[p] = 0xFF
                if p + 1 < len(f) and rng.integers(2):
                    f[p + 1] = 0xC0 + int(rng.integers(0x40))
        else:
            a, n = int(rng.integers(len(f))), int(rng.integers(300))
            f = f[:a] + f[a: a + n] + f[a:]
        files.append(bytes(f))
    batch = jpeg.decode_files(files, gpu)
    torch.cuda.synchronize()
    assert set(batch.status.tolist()) <= set(jpeg.STATUS_TEXT)
    n_ok = 0
    for i in range(len(files)):
        img = batch.image(i)
        if img is not None:
            assert img.shape == (int(batch.height[i]), int(batch.width[i]), 3) and img.numel() <= 3 * (1 << 24)
            n_ok += 1
    assert 0 < n_ok < len(files)
    # the device is still healthy and a clean file still decodes to PIL's bytes afterwards
    good = jpeg.decode_files([seeds[0]], gpu)
    assert np.array_equal(good.image(0).cpu().numpy(), pil_rgb(seeds[0]))


def test_cut_short_file_between_same_size_files_does_not_shift_its_neighbours(gpu):
    """a file cut in the middle of its scan keeps an output slot inside its size class (status 10 is known only after the
    decode); groups() must still hand every good image its own pixels, and the corpus embedding route must give the files
    behind it the embedding they get alone (jpeg.py groups())"""
    from domain_rag_amd import jpeg
    rng = np.random.default_rng(5)
    files = [encode(natural_image(rng, 96, 128), quality=85, subsampling=2) for _ in range(7)]
    files += [encode(natural_image(rng, 64, 64), quality=85) for _ in range(3)]
    good = list(files)
    for i in (2, 3, 8):
        files[i] = files[i][: len(files[i]) * 3 // 5]
    batch = jpeg.decode_files(files, gpu)
    assert [int(s) for s in batch.status] == [10 if i in (2, 3, 8) else 0 for i in range(10)]
    seen = []
    for (h, w), idx, imgs in batch.groups():
        for r, i in enumerate(idx.tolist()):
            assert np.array_equal(imgs[r].cpu().numpy(), pil_rgb(good[i])), i
        seen += idx.tolist()
    assert sorted(seen) == [0, 1, 4, 5, 6, 7, 9]
