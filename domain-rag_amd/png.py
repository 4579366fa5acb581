"""uint8 images on the GPU -> PNG files (``drag_png_encode``): the reference's ``image.save(path)`` of its large RGB results
(batch_generate_flux_kshot.py:480, outpainting_updown_sampling_redux.py:1262,1278) without the host's zlib.  SURVEY §8(f)-2.

A PNG is defined by the pixels it decodes to: the files differ from Pillow's bytes (one dynamic-Huffman block of literals
instead of zlib level 6 — within ~10 % of its size on photographic content) and decode to exactly the array handed in.

    files = encode(images)            # uint8 [n, H, W, 3] (or [n, H, W] / [n, H, W, 1] grey) on the device -> list of bytes
    save(images, paths)               # ... written to disk
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .ops import _p, _stream, check

_scratch: dict = {}


def _buffers(device, ws_bytes: int, out_bytes: int):
    """scratch + output buffers, reused per device and grown geometrically (a stage writes thousands of same-size images)"""
    key = str(device)
    ws, out = _scratch.get(key, (None, None))
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty(max(ws_bytes, 2 * (ws.numel() if ws is not None else 0)), dtype=torch.uint8, device=device)
    if out is None or out.numel() < out_bytes:
        out = torch.empty(max(out_bytes, 2 * (out.numel() if out is not None else 0)), dtype=torch.uint8, device=device)
    _scratch[key] = (ws, out)
    return ws, out


def encode(images: torch.Tensor) -> list:
    """``images``: uint8, on the GPU: ``[n, H, W, 3]`` / ``[H, W, 3]`` (RGB files) or ``[n, H, W, 1]`` / ``[n, H, W]`` (8-bit grey
    files).  A 3-D tensor whose last dimension is 1 or 3 is ONE image; any other 3-D tensor is a batch of grey images.
    Returns one ``bytes`` object (a whole .png file) per image."""
    lib = _lib.load()
    if images.dtype != torch.uint8:
        raise TypeError("png.encode: uint8 images expected")
    if images.device.type != "cuda":
        raise RuntimeError("png.encode: the PNG encoder is a GPU path (domain-rag_amd has no CPU fallback)")
    if images.dim() == 3 and images.shape[-1] in (1, 3):
        images = images[None]
    if images.dim() == 3:
        images = images[..., None]
    if images.dim() != 4 or images.shape[-1] not in (1, 3):
        raise ValueError(f"png.encode: [n, H, W, 3] or [n, H, W, 1] expected, got {tuple(images.shape)}")
    images = images.contiguous()
    n, H, W, C = (int(v) for v in images.shape)
    if n == 0:
        return []
    ws_bytes, stride = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.drag_png_plan(n, H, W, C, ctypes.byref(ws_bytes), ctypes.byref(stride)), "drag_png_plan")
    ws, out = _buffers(images.device, ws_bytes.value + 256, n * stride.value)
    pad = (-ws.data_ptr()) % 256
    sizes = torch.empty(n, dtype=torch.int64, device=images.device)
    check(lib.drag_png_encode(_p(images), n, H, W, C, ws.data_ptr() + pad, ws.numel() - pad, _p(out), stride.value, _p(sizes),
                              _stream()), "drag_png_encode")
    host_sizes = sizes.cpu().tolist()                      # the one synchronisation
    files = []
    for i, sz in enumerate(host_sizes):
        files.append(out[i * stride.value: i * stride.value + sz].cpu().numpy().tobytes())
    return files


def save(images: torch.Tensor, paths) -> None:
    files = encode(images)
    if len(files) != len(paths):
        raise ValueError("png.save: one path per image")
    for data, path in zip(files, paths):
        with open(path, "wb") as f:
            f.write(data)
