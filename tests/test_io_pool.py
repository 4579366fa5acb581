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
