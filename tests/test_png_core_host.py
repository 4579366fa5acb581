"""csrc/png_core.h compiled for the HOST (tests/helpers/png_host.cpp): files written by the serial composition of the functions
the GPU kernels run must be read back by PIL (libpng + zlib, i.e. an independent reader that verifies the CRC-32, the Adler-32
and the Huffman tables) as exactly the input pixels."""
import ctypes
import io
import os
import subprocess
import zlib

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("png_host") / "libpng_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, os.path.join(ROOT, "tests", "helpers", "png_host.cpp")],
                   check=True)
    lib = ctypes.CDLL(so)
    lib.png_host_encode.restype = ctypes.c_int64
    lib.png_host_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    lib.png_host_code_lengths.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return lib


def _encode(lib, arr: np.ndarray, piece: int = 256) -> bytes:
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape[:2]
    c = 1 if arr.ndim == 2 else arr.shape[2]
    cap = 2 * arr.size + 2 * h + 4096
    out = np.zeros(cap, dtype=np.uint8)
    n = lib.png_host_encode(arr.ctypes.data, h, w, c, out.ctypes.data, cap, piece)
    assert n > 0
    return out[:n].tobytes()


def _photo_like(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], axis=-1)
    return np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(1, 1, 3), (1, 7, 3), (5, 1, 3), (37, 53, 3), (64, 64, 3), (200, 333, 3), (17, 19), (1, 1), (128, 96)])
def test_files_decode_to_the_input(host, shape):
    rng = np.random.default_rng(sum(shape))
    for kind in ("noise", "photo", "flat", "ramp"):
        if kind == "noise":
            a = rng.integers(0, 256, shape, dtype=np.uint8)
        elif kind == "photo":
            a = _photo_like(shape[0], shape[1], 3)
            a = a if len(shape) == 3 else a[..., 0].copy()
        elif kind == "flat":
            a = np.full(shape, 77, dtype=np.uint8)
        else:
            a = (np.arange(int(np.prod(shape))) % 251).astype(np.uint8).reshape(shape)
        data = _encode(host, a)
        im = Image.open(io.BytesIO(data))
        im.load()
        assert im.mode == ("RGB" if len(shape) == 3 else "L") and im.size == (shape[1], shape[0])
        assert np.array_equal(np.asarray(im), a), kind


def test_checksums_and_piecewise_crc(host):
    a = _photo_like(120, 160, 5)
    ref = _encode(host, a, piece=1 << 30)                      # one CRC piece
    for piece in (1, 3, 64, 256, 1000, 4096):
        assert _encode(host, a, piece=piece) == ref, piece        # crc(A || B) by the GF(2) combine
    # walk the chunks like a reader: IHDR, IDAT, IEND with their CRCs; the zlib stream inflates to the filtered rows
    assert ref[:8] == b"\x89PNG\r\n\x1a\n"
    pos, kinds = 8, []
    idat = b""
    while pos < len(ref):
        n = int.from_bytes(ref[pos:pos + 4], "big"); typ = ref[pos + 4:pos + 8]; body = ref[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == int.from_bytes(ref[pos + 8 + n:pos + 12 + n], "big"), typ
        kinds.append(typ)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    assert kinds == [b"IHDR", b"IDAT", b"IEND"]
    raw = zlib.decompress(idat)                                  # verifies the Adler-32
    assert len(raw) == 120 * (1 + 160 * 3)
    assert set(raw[::1 + 160 * 3]) <= {0, 1, 2, 3, 4}           # the filter bytes


def test_compresses_photographic_content_like_huffman_only_zlib(host):
    a = _photo_like(256, 256, 9)
    mine = len(_encode(host, a))
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="PNG"); pil6 = len(buf.getvalue())
    assert mine < 0.8 * a.size, (mine, a.size)                  # entropy coding of the filtered rows pays
    assert mine < 1.25 * pil6, (mine, pil6)                      # and is not far from zlib level 6 on such content


def test_code_lengths_are_limited_to_15_bits_and_complete(host):
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for counts in ([fib[i] if i < 40 else 0 for i in range(257)],           # the depth-39 worst case
                   [1] * 257, [0] * 256 + [1], [5] + [0] * 255 + [1], [2 ** 31] * 3 + [0] * 253 + [1],
                   list(np.random.default_rng(1).integers(0, 1000, 257))):
        c = np.array(counts, dtype=np.uint32)
        c[256] = max(c[256], 1)
        ln = np.zeros(257, dtype=np.uint8)
        host.png_host_code_lengths(c.ctypes.data, ln.ctypes.data)
        used = c > 0
        assert (ln[used] >= 1).all() and (ln[~used] == 0).all() and ln.max() <= 15
        if used.sum() > 1:
            assert abs(sum(2.0 ** -int(v) for v in ln[used]) - 1.0) < 1e-12           # Kraft equality: a complete code
