"""oracle/resize.py — CPU restatement of Pillow's 8-bit separable resample (``Image.resize`` for mode RGB / L with
BILINEAR, BICUBIC or LANCZOS), the resize every image on the retrieval path goes through:
  openai-CLIP ``_transform`` Resize(224, BICUBIC) + CenterCrop        retrieval/clip100_resnet_style_all_shots.py:209,171
  SiglipImageProcessor resize(384x384, BICUBIC) inside FluxPriorReduxPipeline   batch_…:459-465, outpainting_…:1237-1243
TEST INFRASTRUCTURE ONLY.

PINNED: unlike the rest of oracle/, the dependency that holds this arithmetic (pillow, requirements.txt:44 pins
11.2.1; 12.2.0 is installed here, same resampler) is importable, so tests/test_oracle_resize.py checks this
restatement bit-for-bit against ``PIL.Image.resize`` itself, and the GPU tests check the HIP kernel against PIL.

Algorithm (published Pillow source, libImaging/Resample.c): per output index the filter window
[center - support, center + support) with support = filter_support * max(scale, 1); double-precision weights,
normalised to sum 1, converted to 22-bit fixed point with round-half-away; each pass accumulates
``(1 << 21) + sum(pixel * k)`` in int32, shifts right by 22 and clamps to [0, 255].  Horizontal pass first (only the
rows the vertical pass will read), then vertical, each rounding to uint8.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


def precompute_coeffs(in_size: int, out_size: int, filt: str):
    """-> (bounds int [out,2] = (xmin, count), kk int32 [out, ksize]) for the whole axis (box = full image)"""
    f, fsupport = FILTERS[filt]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int, out_size: int, offset: int = 0) -> np.ndarray:
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, dtype=np.uint8)
    src = img.astype(np.int64)
    for i in range(out_size):
        x0, n = int(bounds[i, 0]) - offset, int(bounds[i, 1])
        k = kk[i, :n].astype(np.int64)
        if axis == 1:
            acc = (src[:, x0:x0 + n] * k[None, :, None]).sum(axis=1)
        else:
            acc = (src[x0:x0 + n] * k[:, None, None]).sum(axis=0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        if axis == 1:
            out[:, i] = np.clip(acc, 0, 255)
        else:
            out[i] = np.clip(acc, 0, 255)
    return out


def resize_u8(img: np.ndarray, out_w: int, out_h: int, filt: str = "bicubic") -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], as ``Image.fromarray(img).resize((out_w, out_h), FILTER)``"""
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, _ = img.shape
    need_h, need_v = out_w != W, out_h != H
    bv = kv = None
    if need_v:
        bv, kv = precompute_coeffs(H, out_h, filt)
    cur, first = img, 0
    if need_h:
        bh, kh = precompute_coeffs(W, out_w, filt)
        if need_v:   # only the rows the vertical pass reads
            first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
            cur = cur[first:last]
        cur = _pass(cur, bh, kh, 1, out_w)
    if need_v:
        cur = _pass(cur, bv, kv, 0, out_h, offset=first if need_h else 0)
    out = cur.copy() if cur is img else cur
    return out[:, :, 0] if squeeze else out
