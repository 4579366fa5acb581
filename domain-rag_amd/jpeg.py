"""JPEG files -> RGB pixels on the GPU (``drag_jpeg_parse`` / ``drag_jpeg_decode_rgb``), byte-identical to
``PIL.Image.open(f).convert("RGB")`` — the decode the reference pays per corpus image before CLIP's preprocess
(retrieval/clip100_resnet_style_all_shots.py:270-281).  SURVEY §8(f)-2.

The host's part is reading files: a batch travels as ONE byte blob + offsets; markers are parsed on the device, the only
read-back is the per-file descriptor (size, sampling, status) needed to size the outputs.  Files the device path does not
cover (CMYK, arithmetic coding, unusual sampling — ``status != 0``; progressive files ARE decoded on the device since round 3) are reported, not guessed: the caller decodes those few with
PIL (same bytes by definition).

    batch = decode_files([bytes, ...], device)      # -> DecodedBatch
    batch.image(i)                                   # uint8 [H, W, 3] view on the device, or None if status[i] != 0
    for (h, w), idx, imgs in batch.groups():         # same-size images as one dense [m, h, w, 3] tensor (for the resample kernel)
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream, check

INFO_WORDS = 48
STATUS_TEXT = {0: "ok", 1: "not a JPEG", 2: "truncated header", 3: "arithmetic / lossless / hierarchical frame type", 4: "not 8-bit",
               5: "not grey or YCbCr", 6: "sampling other than 4:4:4 / 4:2:2 / 4:2:0", 7: "multi-scan", 8: "table problem",
               9: "chroma at most 2 samples wide", 11: "more than 2^24 pixels", 10: "entropy-coded data does not end at EOI (cut short / trailing data / damaged), or a progressive file whose scans stop early"}


class DecodedBatch:
    def __init__(self, info: np.ndarray, out: torch.Tensor, out_off: np.ndarray, order: np.ndarray):
        self.info = info                    # int32 [n, 48] (drag_jpeg_info words)
        self.status = info[:, 0].copy()
        self.width, self.height = info[:, 1].copy(), info[:, 2].copy()
        self._out, self._off, self._order = out, out_off, order

    def __len__(self) -> int:
        return len(self.status)

    def image(self, i: int):
        if self.status[i] != 0:
            return None
        h, w = int(self.height[i]), int(self.width[i])
        o = int(self._off[i])
        return self._out[o: o + h * w * 3].view(h, w, 3)

    def groups(self):
        """((h, w), indices int64 array, uint8 [m, h, w, 3] dense device tensor) per run of same-size images: the output buffer
        is laid out size class by size class, so a class is one contiguous slab — EXCEPT where a file was planned a slot and
        then rejected (status 10: its scan does not end at EOI, known only after the decode).  Such a slot stays in the middle
        of its class, so a run also ends where the next good image does not start right behind the previous one; a class
        with rejected files comes out as several groups, every row of every group is exactly ``image(idx[r])``."""
        ok = self._order[self.status[self._order] == 0]
        i = 0
        while i < len(ok):
            h, w = int(self.height[ok[i]]), int(self.width[ok[i]])
            step = h * w * 3
            j = i + 1
            while (j < len(ok) and int(self.height[ok[j]]) == h and int(self.width[ok[j]]) == w
                   and int(self._off[ok[j]]) == int(self._off[ok[j - 1]]) + step):
                j += 1
            idx = ok[i:j]
            o = int(self._off[idx[0]])
            yield (h, w), idx, self._out[o: o + (j - i) * h * w * 3].view(j - i, h, w, 3)
            i = j


_staging: dict = {}
PAD = 64            # readable bytes after the last file (the bit reader's read-ahead window looks up to 32 bytes past a file)


def _staging_buffer(device, nbytes: int, slot: int = 0) -> torch.Tensor:
    """pinned host buffer of at least nbytes (reused per (device, slot), grown geometrically)"""
    key = (str(device), slot)
    buf = _staging.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 2 * (buf.numel() if buf is not None else 0)), dtype=torch.uint8).pin_memory()
        _staging[key] = buf
    return buf


class StagedFiles:
    """a batch of files sitting in a pinned host buffer, ready for ONE upload: ``buf[:offsets[-1] + PAD]``; ``errors`` maps the
    index of a file that could not be read to the exception (its slot is empty: offsets[i] == offsets[i + 1])"""

    def __init__(self, buf: torch.Tensor, offsets: np.ndarray, errors: dict):
        self.buf, self.offsets, self.errors = buf, offsets, errors
        self._dev = None                    # the uploaded blob (made once: the header parse and the decode share it)

    def device_blob(self, device) -> torch.Tensor:
        if self._dev is None or self._dev.device != torch.device(device):
            self._dev = self.buf[: int(self.offsets[-1]) + PAD].to(device, non_blocking=True)
        return self._dev

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def file_bytes(self, i: int) -> bytes:
        return self.buf.numpy()[self.offsets[i]: self.offsets[i + 1]].tobytes()

    def slice(self, a: int, b: int) -> "StagedFiles":
        """files [a, b) as a batch of their own (a view of the same pinned buffer; the PAD bytes behind it are the next file's)"""
        o = int(self.offsets[a])
        sub = StagedFiles(self.buf[o:], self.offsets[a: b + 1] - o, {k - a: v for k, v in self.errors.items() if a <= k < b})
        if self._dev is not None:           # already uploaded: the piece is a view of the parent's device blob
            sub._dev = self._dev[o: int(self.offsets[b]) + PAD]
        return sub


def parse_pixels(staged: "StagedFiles", device="cuda") -> np.ndarray:
    """width * height per file as the device parser reads the headers (0 for files it will not decode): what a caller needs to
    cut a batch so that its decode buffers (about 7.5 bytes per pixel: RGB out, int16 coefficients, sample planes) fit a budget"""
    lib = _lib.load()
    n = len(staged)
    device = torch.device(device)
    data = staged.device_blob(device)
    d_off = torch.from_numpy(np.ascontiguousarray(staged.offsets)).to(device)
    d_info = torch.empty((n, INFO_WORDS), dtype=torch.int32, device=device)
    check(lib.drag_jpeg_parse(_p(data), _p(d_off), n, _p(d_info), _stream()), "drag_jpeg_parse")
    info = d_info.cpu().numpy()
    return np.where(info[:, 0] == 0, info[:, 1].astype(np.int64) * info[:, 2].astype(np.int64), 0)


def budget_bounds(pixels: np.ndarray, max_pixels: int) -> list:
    """[(a, b)] consecutive index ranges whose pixel sums stay within ``max_pixels`` (a single file above the budget gets a range
    of its own: the parser already refuses anything above 2^24 pixels)"""
    bounds, a, acc = [], 0, 0
    for i, px in enumerate(pixels.tolist()):
        if i > a and acc + px > max_pixels:
            bounds.append((a, i))
            a, acc = i, 0
        acc += px
    if a < len(pixels):
        bounds.append((a, len(pixels)))
    return bounds


def stage_paths(paths, device="cuda", slot: int = 0, threads: int = 32) -> StagedFiles:
    """read files straight into the pinned staging buffer (no bytes objects, no concatenation pass) with the library's native
    reader threads (``drag_file_sizes`` / ``drag_read_files``): sizes first, then every file at its offset"""
    import os
    lib = _lib.load()
    n = len(paths)
    enc = [os.fsencode(p) for p in paths]
    arr = (ctypes.c_char_p * n)(*enc)
    sizes = np.empty(n, dtype=np.int64)
    check(lib.drag_file_sizes(arr, n, sizes.ctypes.data, threads), "drag_file_sizes")
    errors = {int(i): OSError(int(-sizes[i]), os.strerror(int(-sizes[i])), paths[i]) for i in np.nonzero(sizes < 0)[0]}
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.maximum(sizes, 0), out=offsets[1:])
    total = int(offsets[-1])
    buf = _staging_buffer(device, total + PAD, slot)
    host = buf.numpy()
    status = np.zeros(n, dtype=np.int32)
    check(lib.drag_read_files(arr, n, host.ctypes.data, offsets.ctypes.data, status.ctypes.data, threads), "drag_read_files")
    for i in np.nonzero(status != 0)[0].tolist():
        code = int(status[i])
        errors[i] = OSError(code, os.strerror(code), paths[i]) if code > 0 else OSError(f"short read: {paths[i]}")
        host[offsets[i]: offsets[i + 1]] = 0          # not a JPEG any more: the parse kernel reports it, nothing decodes it
    host[total: total + PAD] = 0
    return StagedFiles(buf, offsets, errors)


def _upload(blobs, device):
    """concatenate bytes objects into the pinned staging buffer and upload once"""
    sizes = np.fromiter((len(b) for b in blobs), dtype=np.int64, count=len(blobs))
    offsets = np.zeros(len(blobs) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    total = int(offsets[-1])
    buf = _staging_buffer(device, total + PAD)
    host = buf.numpy()
    for b, o in zip(blobs, offsets[:-1]):
        host[o: o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    host[total: total + PAD] = 0
    data = buf[: total + PAD].to(device, non_blocking=True)
    return data, offsets


def decode_files(blobs, device="cuda", check_scan: bool = True) -> DecodedBatch:
    """``blobs``: list of bytes objects (whole files), or a ``StagedFiles`` (``stage_paths``).  All arithmetic runs in
    libdomainrag_hip.so.  ``check_scan``: read the per-file end-of-scan flags back (one more synchronisation) and mark files
    whose entropy data does not end at EOI."""
    lib = _lib.load()
    n = len(blobs)
    if n == 0:
        raise ValueError("decode_files: empty batch")
    if n > 65535:
        raise ValueError("decode_files: at most 65535 files per batch")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("decode_files: the JPEG decoder is a GPU path (domain-rag_amd has no CPU fallback)")
    if isinstance(blobs, StagedFiles):
        offsets = blobs.offsets
        data = blobs.device_blob(device)
    else:
        data, offsets = _upload(blobs, device)
    d_off = torch.from_numpy(np.ascontiguousarray(offsets)).to(device)
    d_info = torch.empty((n, INFO_WORDS), dtype=torch.int32, device=device)
    check(lib.drag_jpeg_parse(_p(data), _p(d_off), n, _p(d_info), _stream()), "drag_jpeg_parse")
    info = d_info.cpu().numpy()                       # the one synchronisation: sizes are needed to plan the outputs
    ok = info[:, 0] == 0
    w, h, ncomp = info[:, 1].astype(np.int64), info[:, 2].astype(np.int64), info[:, 3]
    hs, vs = info[:, 4:7].astype(np.int64), info[:, 7:10].astype(np.int64)
    mx, my = info[:, 21].astype(np.int64), info[:, 22].astype(np.int64)
    blocks = np.zeros(n, dtype=np.int64)
    for c in range(3):
        blocks += np.where(ok & (ncomp > c), mx * hs[:, c] * my * vs[:, c], 0)
    pixels = np.where(ok, w * h, 0)
    # outputs grouped by size class (stable within a class), so same-size images form one dense slab
    order = np.lexsort((np.arange(n), w, h)).astype(np.int64)
    plan = np.zeros((n, 3), dtype=np.int64)
    co = po = oo = 0
    for i in order:
        plan[i] = (co, po, oo)
        co += blocks[i] * 64
        po += blocks[i] * 64
        oo += pixels[i] * 3
    out = torch.empty(max(oo, 1), dtype=torch.uint8, device=device)
    if not ok.any():
        return DecodedBatch(info, out, plan[:, 2], order)
    coef = torch.empty(max(co, 1), dtype=torch.int16, device=device)
    planes = torch.empty(max(po, 1), dtype=torch.uint8, device=device)
    qtab = torch.empty((n, 3, 64), dtype=torch.int16, device=device)
    d_plan = torch.from_numpy(plan).to(device)
    scan = torch.empty(n, dtype=torch.int32, device=device)
    check(lib.drag_jpeg_decode_rgb(_p(data), _p(d_off), _p(d_info), _p(d_plan), n, int(blocks.max()), int(pixels.max()),
                                   _p(coef), coef.numel() * 2, _p(planes), _p(qtab), _p(out), _p(scan), _stream()), "drag_jpeg_decode_rgb")
    if check_scan:      # a scan that does not end at EOI: let libjpeg / PIL decide what the file means (status 10)
        info[:, 0] = np.where((info[:, 0] == 0) & (scan.cpu().numpy() != 0), 10, info[:, 0])
    return DecodedBatch(info, out, plan[:, 2], order)


def info_dict(row: np.ndarray) -> dict:
    """one descriptor row as a dict (field order of ``drag_jpeg_info``)"""
    r = [int(v) for v in row]
    return dict(status=r[0], width=r[1], height=r[2], ncomp=r[3], hs=r[4:7], vs=r[7:10], tq=r[10:13], td=r[13:16], ta=r[16:19],
                hmax=r[19], vmax=r[20], mcus_x=r[21], mcus_y=r[22], restart_interval=r[23], scan_off=r[24], dqt_off=r[25:29],
                dqt_16=r[29:33], dht_off=r[33:41], progressive=r[41], cid=r[42:45])
