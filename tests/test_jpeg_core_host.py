"""csrc/jpeg_core.h compiled for the HOST (tests/helpers/jpeg_host.cpp) against PIL's own decode: the Huffman decoding, ISLOW
IDCT, fancy upsampling and YCbCr -> RGB arithmetic the gfx950 kernels run must reproduce Image.open(f).convert("RGB") byte for
byte (retrieval/clip100_resnet_style_all_shots.py:270-281 decodes every corpus image that way), or top-k stops being bit-exact.
The GPU kernels themselves are compared with PIL in tests/test_gpu_jpeg.py; this file checks the shared arithmetic where there
is no GPU."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("jpeg_host") / "libjpeg_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, os.path.join(ROOT, "tests", "helpers", "jpeg_host.cpp")],
                   check=True)
    lib = ctypes.CDLL(so)
    lib.jpeg_host_decode_rgb.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p]
    lib.jpeg_host_info.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p]
    return lib


def natural_image(rng, h, w):
    """smooth colour fields + noise: exercises long and short Huffman codes, EOB and ZRL runs"""
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16) + rng.integers(-20, 20, (h, w, 3))
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


def encode(im, **kw):
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, im.size[0] * im.size[1] * 4)     # optimize=True at quality 100 needs one big buffer
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    return bio.getvalue()


def decode_both(host, data):
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    out = np.zeros_like(ref)
    st = host.jpeg_host_decode_rgb(data, len(data), out.ctypes.data)
    return st, out, ref


@pytest.mark.parametrize("size", [(640, 480), (33, 17), (64, 64), (101, 77), (8, 8), (17, 40), (500, 375), (5, 3), (16, 1)])
def test_baseline_decode_is_byte_identical_to_pil(host, size):
    rng = np.random.default_rng(size[0] * 1000 + size[1])
    w, h = size
    for sub in (0, 1, 2):
        for q in (10, 30, 75, 95, 100):
            for kw in ({}, {"optimize": True}):
                st, out, ref = decode_both(host, encode(natural_image(rng, h, w), quality=q, subsampling=sub, **kw))
                assert st == 0 and np.array_equal(out, ref), (size, sub, q, kw)
    for q in (30, 90):
        st, out, ref = decode_both(host, encode(natural_image(rng, h, w).convert("L"), quality=q))
        assert st == 0 and np.array_equal(out, ref), (size, "L", q)


def test_noise_images_and_saturation(host):
    """pure noise and hard edges drive IDCT outputs past [0, 255]: the clamp must match the library's"""
    rng = np.random.default_rng(1)
    for q in (5, 50, 98):
        for sub in (0, 2):
            im = Image.fromarray(rng.integers(0, 256, (96, 120, 3), dtype=np.uint8))
            st, out, ref = decode_both(host, encode(im, quality=q, subsampling=sub))
            assert st == 0 and np.array_equal(out, ref), (q, sub)
    a = np.zeros((64, 64, 3), np.uint8); a[::2, ::3] = 255; a[5:20, 30:] = (255, 0, 0)
    st, out, ref = decode_both(host, encode(Image.fromarray(a), quality=85, subsampling=2))
    assert st == 0 and np.array_equal(out, ref)


def test_restart_intervals(host):
    rng = np.random.default_rng(2)
    im = natural_image(rng, 90, 130)
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}, {"restart_marker_blocks": 1}):
        for sub in (0, 2):
            data = encode(im, quality=80, subsampling=sub, **kw)
            if b"\xff\xdd" not in data:
                pytest.skip("this Pillow does not write restart markers")
            st, out, ref = decode_both(host, data)
            assert st == 0 and np.array_equal(out, ref), (kw, sub)


def test_unsupported_variants_are_reported_not_guessed(host):
    rng = np.random.default_rng(3)
    im = natural_image(rng, 40, 56)
    info = (ctypes.c_int32 * 48)()
    prog = bytearray(encode(im, quality=80, progressive=True))
    i = prog.index(b"\xff\xc2"); prog[i + 1] = 0xCA                     # SOF10: progressive, ARITHMETIC coding
    cases = {"arithmetic progressive": (bytes(prog), 3),
             "cmyk": (encode(im.convert("CMYK"), quality=80), 5),
             "not a jpeg": (b"\x89PNG\r\n\x1a\n" + b"\0" * 32, 1),
             "truncated header": (encode(im, quality=80)[:40], 2),
             "narrow subsampled": (encode(natural_image(rng, 9, 4), quality=80, subsampling=2), 9)}
    for name, (data, want) in cases.items():
        assert host.jpeg_host_info(data, len(data), info) == want, name
    ok = encode(im, quality=80, subsampling=1)
    assert host.jpeg_host_info(ok, len(ok), info) == 0
    assert (info[1], info[2], info[3]) == (56, 40, 3) and (info[4], info[7]) == (2, 1)      # width, height, ncomp; luma 2x1


@pytest.mark.parametrize("size", [(64, 48), (33, 17), (101, 77), (8, 8), (17, 40), (5, 3), (16, 1), (200, 150)])
def test_progressive_decode_is_byte_identical_to_pil(host, size):
    """round 3: SOF2 files — libjpeg's default progression (interleaved DC first, per-component AC first scans with spectral
    selection, AC and DC refinement scans), every subsampling, grey, restart intervals, optimised tables — decode to PIL's bytes"""
    rng = np.random.default_rng(size[0] * 131 + size[1])
    w, h = size
    im = natural_image(rng, h, w)
    cases = [dict(quality=q, subsampling=sub) for q in (30, 75, 92, 100) for sub in (0, 1, 2)]
    cases += [dict(quality=85, subsampling=2, optimize=True), dict(quality=60, subsampling=1, restart_marker_blocks=2),
              dict(quality=90, subsampling=0, restart_marker_rows=1)]
    info = (ctypes.c_int32 * 48)()
    for kw in cases:
        if w <= 4 and kw.get("subsampling", 0) != 0:
            continue                                                        # (status 9: chroma too narrow for the triangle filters)
        data = encode(im, progressive=True, **kw)
        assert host.jpeg_host_info(data, len(data), info) == 0 and info[41] == 1, kw      # parsed as progressive
        st, out, ref = decode_both(host, data)
        assert st == 0, (kw, st)
        assert np.array_equal(out, ref), (kw, int(np.abs(out.astype(int) - ref).max()))
    grey = encode(im.convert("L"), quality=80, progressive=True)
    st, out, ref = decode_both(host, grey)
    assert st == 0 and np.array_equal(out, ref)
    noise = encode(Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)), quality=97, progressive=True, subsampling=0)
    st, out, ref = decode_both(host, noise)
    assert st == 0 and np.array_equal(out, ref)


def test_progressive_file_cut_after_some_scans_is_reported(host):
    """a progressive file whose later scans are missing (EOI spliced in after the third scan): libjpeg would smooth the blocks, so the
    decoder reports it (102) instead of producing different pixels; one cut inside a scan is reported as damaged (101)"""
    rng = np.random.default_rng(8)
    data = encode(natural_image(rng, 48, 64), quality=85, progressive=True, subsampling=2)
    sos = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    assert len(sos) >= 6
    early = data[:sos[3]] + b"\xff\xd9"
    out = np.zeros((48, 64, 3), np.uint8)
    assert host.jpeg_host_decode_rgb(early, len(early), out.ctypes.data) == 102
    cut = data[:sos[2] + 40]
    assert host.jpeg_host_decode_rgb(cut, len(cut), out.ctypes.data) == 101


def test_truncated_scan_does_not_crash_and_keeps_the_decoded_part(host):
    """a file cut inside the entropy data: libjpeg pads with zero bits; whatever it is, the decoder must stay inside the buffer"""
    rng = np.random.default_rng(4)
    data = encode(natural_image(rng, 64, 64), quality=80, subsampling=0)
    cut = data[: len(data) * 2 // 3]
    out = np.zeros((64, 64, 3), np.uint8)
    assert host.jpeg_host_decode_rgb(cut, len(cut), out.ctypes.data) == 0
    full = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(out[:16], full[:16])          # the first MCU rows come from intact data


def test_mutation_fuzz_under_address_sanitizer(tmp_path):
    """whatever the bytes are, the decoder's shared arithmetic stays inside its buffers and away from undefined behaviour
    (on the GPU an out-of-bounds access is a memory fault, not an exception): tests/helpers/jpeg_fuzz.cpp mutates valid files
    (byte flips, header damage, truncation, stray markers, duplicated slices) under -fsanitize=address,undefined"""
    exe = str(tmp_path / "jpeg_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-o", exe,
                        os.path.join(ROOT, "tests", "helpers", "jpeg_fuzz.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(7)
    seeds = []
    for k, (w, h) in enumerate([(64, 48), (33, 17), (120, 90)]):
        for sub in (0, 1, 2):
            for j, kw in enumerate(({}, {"optimize": True}, {"restart_marker_blocks": 2})):
                p = tmp_path / f"s{k}{sub}{j}.jpg"
                p.write_bytes(encode(natural_image(rng, h, w), quality=int(rng.integers(20, 98)), subsampling=sub, **kw))
                seeds.append(str(p))
        p = tmp_path / f"g{k}.jpg"
        p.write_bytes(encode(natural_image(rng, h, w).convert("L"), quality=70))
        seeds.append(str(p))
        for sub in (0, 2):                                     # progressive files: the scan walker and the refinement passes
            for j, kw in enumerate(({}, {"restart_marker_blocks": 3})):
                p = tmp_path / f"p{k}{sub}{j}.jpg"
                p.write_bytes(encode(natural_image(rng, h, w), quality=int(rng.integers(20, 98)), subsampling=sub, progressive=True, **kw))
                seeds.append(str(p))
    r = subprocess.run([exe, "6000"] + seeds, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "iterations 6000" in r.stdout
    dec = int(r.stdout.split("decoded")[1].split(",")[0])
    assert dec > 500, r.stdout                                # (mutations that still decode: the arithmetic paths really ran)
