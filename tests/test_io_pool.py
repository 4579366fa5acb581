"""Background image writer: the files must be exactly what Image.save writes, failures must come back from flush()."""
import io
import os
import time

import numpy as np
from PIL import Image


def _pictures():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:120, 0:200]
    rgb = np.clip(((np.sin(xx / 17.0) + np.cos(yy / 23.0)) * 60 + 128)[..., None] + rng.normal(0, 9, (120, 200, 3)), 0, 255).astype(np.uint8)
    pics = {"rgb.png": Image.fromarray(rgb), "gray.png": Image.fromarray(rgb[..., 0]), "rgb.jpg": Image.fromarray(rgb),
            "rgba.png": Image.fromarray(np.dstack([rgb, rgb[..., :1]]))}
    with_icc = Image.fromarray(rgb[::-1].copy())
    with_icc.info["icc_profile"] = b"fake-profile-bytes" * 8
    pics["icc.png"] = with_icc
    return pics


def test_async_files_equal_inline_files(tmp_path):
    from domain_rag_amd.io_pool import ImageWriter
    pics = _pictures()
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    for name, im in pics.items():
        im.save(tmp_path / "a" / name)
    w = ImageWriter(workers=2)
    try:
        for rep in range(3):                                   # more jobs than workers, interleaved
            for name, im in pics.items():
                w.save(im, str(tmp_path / "b" / (f"{rep}_" + name)))
        assert w.flush() == []
        for rep in range(3):
            for name in pics:
                assert (tmp_path / "b" / (f"{rep}_" + name)).read_bytes() == (tmp_path / "a" / name).read_bytes(), name
        # save kwargs travel; a palette image is written inline (mode not offloaded) — same bytes again
        w.save(pics["rgb.jpg"], str(tmp_path / "b" / "q.jpg"), quality=85)
        pal = pics["rgb.png"].convert("P")
        w.save(pal, str(tmp_path / "b" / "pal.png"))
        assert w.flush() == []
        ref = io.BytesIO(); pics["rgb.jpg"].save(ref, format="JPEG", quality=85)
        assert (tmp_path / "b" / "q.jpg").read_bytes() == ref.getvalue()
        ref = io.BytesIO(); pal.save(ref, format="PNG")
        assert (tmp_path / "b" / "pal.png").read_bytes() == ref.getvalue()
        # a failing path is reported by flush(), the others still land
        w.save(pics["rgb.png"], str(tmp_path / "missing_dir" / "x.png"))
        w.save(pics["gray.png"], str(tmp_path / "b" / "after.png"))
        errs = w.flush()
        assert len(errs) == 1 and errs[0][0].endswith("missing_dir/x.png") and (tmp_path / "b" / "after.png").exists()
        assert w.flush() == []
    finally:
        w.close()


def test_inline_mode_and_returns_before_the_encode(tmp_path):
    from domain_rag_amd.io_pool import ImageWriter
    pics = _pictures()
    w0 = ImageWriter(workers=0)
    w0.save(pics["rgb.png"], str(tmp_path / "inline.png"))
    assert os.path.exists(tmp_path / "inline.png") and w0.flush() == []
    big = Image.fromarray(np.random.default_rng(1).integers(0, 256, (1024, 1024, 3), dtype=np.uint8))
    t = time.perf_counter(); big.save(tmp_path / "ref.png"); inline = time.perf_counter() - t
    w = ImageWriter(workers=1)
    try:
        w.save(pics["gray.png"], str(tmp_path / "warm.png")); w.flush()          # worker start-up outside the timing
        t = time.perf_counter(); w.save(big, str(tmp_path / "bg.png")); queued = time.perf_counter() - t
        assert w.flush() == []
        assert queued < 0.5 * inline, (queued, inline)
        assert (tmp_path / "bg.png").read_bytes() == (tmp_path / "ref.png").read_bytes()
    finally:
        w.close()


def test_clip_decode_pool_matches_host_preprocess(tmp_path):
    """worker processes return exactly the uint8 crop the host-side clip preprocess resizes to, in path order; unreadable
    files come back as failures at their position"""
    import torch
    from domain_rag_amd import retrieval as R
    from domain_rag_amd.io_pool import ClipDecodePool
    rng = np.random.default_rng(2)
    paths = []
    for i, (h, w) in enumerate([(480, 640), (640, 480), (224, 224), (225, 300), (300, 224), (97, 61), (1000, 224), (333, 777)] * 3):
        p = tmp_path / f"{i:03d}.{'png' if i % 3 else 'jpg'}"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    paths.insert(5, str(tmp_path / "missing.jpg"))
    (tmp_path / "broken.jpg").write_bytes(b"not an image")
    paths.insert(11, str(tmp_path / "broken.jpg"))
    gray = tmp_path / "gray.png"
    Image.fromarray(rng.integers(0, 256, (260, 310), dtype=np.uint8)).save(gray)       # L mode -> convert("RGB")
    paths.append(str(gray))
    pool = ClipDecodePool(3)
    try:
        got = list(pool.run(paths))
    finally:
        pool.close()
    assert [g[0] for g in got] == list(range(len(paths)))
    for (k, ok, payload), p in zip(got, paths):
        if "missing" in p or "broken" in p:
            assert not ok and isinstance(payload, str) and payload
            continue
        assert ok and len(payload) == 224 * 224 * 3
        x = R.clip_preprocess(Image.open(p))                                       # float CHW, normalised
        u8 = torch.frombuffer(bytearray(payload), dtype=torch.uint8).view(224, 224, 3)
        back = ((u8.float().div(255.0) - torch.tensor(R.CLIP_MEAN)) / torch.tensor(R.CLIP_STD)).permute(2, 0, 1)
        assert torch.equal(back, x), p
    pool2 = ClipDecodePool(2)
    try:
        assert list(pool2.run([])) == []
    finally:
        pool2.close()
