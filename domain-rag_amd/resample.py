"""PIL-exact image resize on the GPU (``drag_resample_u8``) and the device-side preprocessing of the two vision
towers.  The host computes Pillow's fixed-point coefficient tables (libImaging/Resample.c ``precompute_coeffs`` +
``normalize_coeffs_8bpc``; cached per (in, out, filter)); the kernels apply them — results are bit-identical to
``PIL.Image.resize`` (tests/test_gpu_resample.py checks against PIL itself), so an embedding does not depend on where
the resize ran.

  clip_preprocess_u8   openai-CLIP ``_transform``: Resize(224, BICUBIC) -> CenterCrop(224) as uint8 (ToTensor /
                       Normalize happen inside ``drag_patchify_u8``)         retrieval/clip100_resnet_style_all_shots.py:209,171
  siglip_resize_u8     SiglipImageProcessor: resize to (384, 384) BICUBIC, aspect ignored
                                                          batch_generate_flux_kshot.py:459-465, outpainting_…:1237-1243
"""
from __future__ import annotations

import ctypes
import functools
import math

import numpy as np
import torch

from . import _lib
from .ops import _stream, check

PRECISION_BITS = 32 - 8 - 2
_SUPPORT = {"bilinear": 1.0, "bicubic": 2.0, "lanczos": 3.0}


def _filter(name: str, x: np.ndarray) -> np.ndarray:
    """Pillow's filter functions on a float64 array, same operation order as the C source"""
    if name == "bilinear":
        ax = np.abs(x)
        return np.where(ax < 1.0, 1.0 - ax, 0.0)
    if name == "bicubic":
        a = -0.5
        ax = np.abs(x)
        inner = ((a + 2.0) * ax - (a + 3.0)) * ax * ax + 1
        outer = (((ax - 5) * ax + 8) * ax - 4) * a
        return np.where(ax < 1.0, inner, np.where(ax < 2.0, outer, 0.0))
    if name == "lanczos":      # libm sin per element: numpy's vectorised sin may differ from libm in the last ulp
        def sinc(v):
            if v == 0.0:
                return 1.0
            v = v * math.pi
            return math.sin(v) / v
        flat = [sinc(v) * sinc(v / 3) if -3.0 <= v < 3.0 else 0.0 for v in x.ravel().tolist()]
        return np.asarray(flat, dtype=np.float64).reshape(x.shape)
    raise ValueError(f"unknown filter {name!r}")


@functools.lru_cache(maxsize=4096)
def coeff_tables(in_size: int, out_size: int, filt: str = "bicubic"):
    """(bounds int32 [out, 2] = (first source index, tap count), kk int32 [out, ksize]) — Pillow's tables"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = _SUPPORT[filt] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0.0).astype(np.int64)
    xmax = np.minimum(np.trunc(center + support + 0.5), float(in_size)).astype(np.int64)
    cnt = xmax - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    valid = taps < cnt[:, None]
    w = _filter(filt, ((taps + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where(valid, w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                     # sequential sum, like the C loop
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS)))
    kk = np.where(valid, fixed, 0.0).astype(np.int32)
    bounds = np.stack([xmin, cnt], axis=1).astype(np.int32)
    return bounds, kk


@functools.lru_cache(maxsize=1024)
def _device_tables(in_size: int, out_size: int, filt: str, lo: int, hi: int, device: str):
    b, k = coeff_tables(in_size, out_size, filt)
    return (torch.from_numpy(np.ascontiguousarray(b[lo:hi])).to(device), torch.from_numpy(np.ascontiguousarray(k[lo:hi])).to(device),
            int(k.shape[1]), int(b[lo, 0]), int(b[hi - 1, 0] + b[hi - 1, 1]))


def resize_u8(img: torch.Tensor, out_w: int, out_h: int, filt: str = "bicubic", crop: tuple[int, int, int, int] | None = None,
              out: torch.Tensor | None = None) -> torch.Tensor:
    """``Image.resize((out_w, out_h), FILTER)`` (then ``.crop((left, top, right, bottom))`` if given) on a device uint8
    tensor [H, W, C] or [B, H, W, C] (same-size batch) -> uint8 [(B,) h, w, C]"""
    lib = _lib.load()
    if img.dtype != torch.uint8 or not img.is_cuda:
        raise ValueError("resize_u8: expected a uint8 tensor on the GPU")
    single = img.dim() == 3
    x = (img[None] if single else img).contiguous()
    B, H, W, C = x.shape
    left, top, right, bottom = crop if crop is not None else (0, 0, out_w, out_h)
    if not (0 <= left < right <= out_w and 0 <= top < bottom <= out_h):
        raise ValueError("resize_u8: crop box must lie inside the resized image")
    w, h = right - left, bottom - top
    if out is None:
        out = torch.empty((B, h, w, C), dtype=torch.uint8, device=x.device)
    elif out.dim() == 3:
        out = out[None]
    if tuple(out.shape) != (B, h, w, C) or out.dtype != torch.uint8 or out.stride(3) != 1 or out.stride(2) != C:
        raise ValueError(f"resize_u8: out must be uint8 {(B, h, w, C)} with dense pixels")
    a = _lib.ResampleArgs()
    a.src, a.dst, a.batch, a.channels = x.data_ptr(), out.data_ptr(), B, C
    a.src_h, a.src_w, a.src_image_stride, a.src_row_stride = H, W, H * W * C, W * C
    a.out_h, a.out_w, a.dst_image_stride, a.dst_row_stride = h, w, out.stride(0), out.stride(1)
    keep = []
    dev = str(x.device)
    a.src_col0, a.src_row0 = left, top
    if out_w != W:
        bx, kx, a.ksize_x, _, _ = _device_tables(W, out_w, filt, left, right, dev)
        a.bx, a.kx = bx.data_ptr(), kx.data_ptr()
        keep += [bx, kx]
    r0, r1 = top, bottom
    if out_h != H:
        by, ky, a.ksize_y, r0, r1 = _device_tables(H, out_h, filt, top, bottom, dev)
        a.by, a.ky = by.data_ptr(), ky.data_ptr()
        keep += [by, ky]
    if out_w != W and out_h != H:
        tmp = torch.empty((B, r1 - r0, w, C), dtype=torch.uint8, device=x.device)
        a.tmp, a.tmp_row0, a.tmp_rows = tmp.data_ptr(), r0, r1 - r0
        keep.append(tmp)
    check(lib.drag_resample_u8(ctypes.byref(a), _stream()), "drag_resample_u8")
    return out[0] if single else out


def clip_resize_plan(w: int, h: int, size: int = 224):
    """(resized w, resized h, crop box) of torchvision Resize(size) + CenterCrop(size) as openai-CLIP composes them"""
    # torchvision Resize(int): the short side becomes `size`, the long one int(size * long / short) — TRUNCATED, not rounded
    nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return nw, nh, (left, top, left + size, top + size)


def clip_preprocess_u8(img: torch.Tensor, size: int = 224, out: torch.Tensor | None = None) -> torch.Tensor:
    """device uint8 RGB [H, W, 3] (or same-size batch) -> uint8 [size, size, 3]: Resize(size, BICUBIC) + CenterCrop(size)"""
    H, W = img.shape[-3], img.shape[-2]
    nw, nh, box = clip_resize_plan(W, H, size)
    return resize_u8(img, nw, nh, "bicubic", crop=box, out=out)


def siglip_resize_u8(img: torch.Tensor, size: int = 384, out: torch.Tensor | None = None) -> torch.Tensor:
    return resize_u8(img, size, size, "bicubic", out=out)
