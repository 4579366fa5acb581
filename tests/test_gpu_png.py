"""GPU PNG encoder (csrc/png.hip) through the C ABI: every file must be read back by PIL (libpng + zlib verify the CRC-32, the
Adler-32 and the Huffman tables) as exactly the input pixels, and must equal, byte for byte, the file the SERIAL composition of
the same png_core.h functions writes on the host (tests/helpers/png_host.cpp)."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("png_host") / "libpng_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, os.path.join(ROOT, "tests", "helpers", "png_host.cpp")],
                   check=True)
    lib = ctypes.CDLL(so)
    lib.png_host_encode.restype = ctypes.c_int64
    lib.png_host_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    return lib


def _host_file(lib, a):
    a = np.ascontiguousarray(a)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    cap = 2 * a.size + 2 * h + 4096
    out = np.zeros(cap, dtype=np.uint8)
    n = lib.png_host_encode(a.ctypes.data, h, w, c, out.ctypes.data, cap, 256)
    return out[:n].tobytes()


def _photo(h, w, seed, c=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 37.0 + k) * np.cos(yy / 23.0 - k) for k in range(c)], axis=-1)
    return np.clip(base + rng.normal(0, 6, (h, w, c)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(1, 1, 1, 3), (2, 1, 9, 3), (3, 5, 1, 3), (2, 37, 53, 3), (1, 64, 64, 3), (4, 200, 333, 3),
                                   (2, 17, 19, 1), (1, 1024, 1024, 3), (2, 1024, 1360, 3), (1, 480, 640, 1)])
def test_gpu_files_decode_to_the_input_and_equal_the_serial_writer(gpu, host, shape):
    from domain_rag_amd import png
    n, h, w, c = shape
    rng = np.random.default_rng(h * 7 + w)
    batch = []
    for i in range(n):
        kind = i % 4
        if kind == 0:
            a = _photo(h, w, i, c)
        elif kind == 1:
            a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        elif kind == 2:
            a = np.full((h, w, c), 200, dtype=np.uint8)
        else:
            a = (np.arange(h * w * c) % 251).astype(np.uint8).reshape(h, w, c)
        batch.append(a)
    arr = np.stack(batch)
    files = png.encode(torch.from_numpy(arr).to(gpu))
    assert len(files) == n
    for a, data in zip(batch, files):
        im = Image.open(io.BytesIO(data))
        im.load()
        assert im.size == (w, h) and im.mode == ("RGB" if c == 3 else "L")
        got = np.asarray(im)
        assert np.array_equal(got if c == 3 else got[..., None], a)
        assert data == _host_file(host, a if c == 3 else a[..., 0])


def test_photographic_size_vs_pillow_and_reuse(gpu):
    """entropy coding of the filtered rows: smaller than raw, near zlib level 6 on such content; buffers are reused across calls
    of different sizes without leaking one image's bytes into the next"""
    from domain_rag_amd import png
    a = _photo(1024, 1024, 11)
    data = png.encode(torch.from_numpy(a).to(gpu))[0]
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="PNG")
    assert len(data) < 0.8 * a.size and len(data) < 1.25 * len(buf.getvalue()), (len(data), len(buf.getvalue()), a.size)
    small = _photo(30, 40, 2)
    for _ in range(3):
        d2 = png.encode(torch.from_numpy(small).to(gpu))[0]
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(d2))), small)
    assert png.encode(torch.from_numpy(a).to(gpu))[0] == data            # deterministic


def test_png_argument_errors(gpu, tmp_path):
    from domain_rag_amd import png
    with pytest.raises(TypeError):
        png.encode(torch.zeros((1, 4, 4, 3), device=gpu))
    with pytest.raises(ValueError):
        png.encode(torch.zeros((1, 4, 4, 2), dtype=torch.uint8, device=gpu))
    with pytest.raises(RuntimeError):
        png.encode(torch.zeros((1, 4, 4, 3), dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="2\\^26"):
        png.encode(torch.zeros((1, 9000, 9000, 1), dtype=torch.uint8, device=gpu))
    p = tmp_path / "x.png"
    a = _photo(33, 21, 1)
    png.save(torch.from_numpy(a).to(gpu), [str(p)])
    assert np.array_equal(np.asarray(Image.open(p)), a)
