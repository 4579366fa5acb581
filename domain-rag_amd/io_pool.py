"""Image I/O off the generation thread for the stage CLIs: background PNG / JPEG encoding (ImageWriter) and corpus decoding
(ClipDecodePool, below).

A 1365x1024 PNG costs ~0.3 s of zlib on one core and Pillow holds the GIL while it runs, so the reference's pattern —
generate, ``image.save(...)``, generate — leaves the GPU idle for 10–35 % of a stage-3 sample (2 large PNGs per composite,
`outpainting_…:1262-1283`; 1 per image in stage 2, `batch_…:514`).  ``ImageWriter`` hands the pixels to a few plain worker
processes (``python -c``: PIL only, no torch, no HIP) over pipes and returns at once; the files are byte-for-byte what
``Image.save`` would have written (same encoder, same mode / size / pixels / ICC profile).  ``flush()`` waits for everything
queued so far and returns the list of failed paths — callers flush before anything reads the files back.
"""
from __future__ import annotations

import atexit
import pickle
import queue
import subprocess
import sys
import threading

_WORKER = r"""
import pickle, sys
from PIL import Image
inp, out = sys.stdin.buffer, sys.stdout.buffer
errors = []
while True:
    try:
        msg = pickle.load(inp)
    except EOFError:
        break
    if msg[0] == "sync":
        out.write(pickle.dumps(errors)); out.flush(); errors = []
    else:
        _, path, mode, size, data, kw = msg
        try:
            Image.frombytes(mode, size, data).save(path, **kw)
        except Exception as e:
            errors.append((path, repr(e)))
"""


class _Worker:
    def __init__(self, depth: int):
        self.proc = subprocess.Popen([sys.executable, "-c", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        self.q: queue.Queue = queue.Queue(maxsize=depth)
        self.dead = None
        self.thread = threading.Thread(target=self._pump, daemon=True)
        self.thread.start()

    def _pump(self):                       # the blocking pipe writes happen here (the GIL is released inside write())
        while True:
            msg = self.q.get()
            if msg is None:
                break
            try:
                if isinstance(msg, threading.Event):
                    msg.set()
                    continue
                pickle.dump(msg, self.proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
                self.proc.stdin.flush()
            except Exception as e:         # worker died: remember, keep draining so producers never block forever
                self.dead = repr(e)

    def sync(self):
        ev = threading.Event()
        self.q.put(ev)
        ev.wait()                          # everything queued before has been written to the pipe
        if self.dead:
            return [("<worker>", self.dead)]
        try:
            pickle.dump(("sync",), self.proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
            self.proc.stdin.flush()
            return pickle.load(self.proc.stdout)
        except Exception as e:
            self.dead = repr(e)
            return [("<worker>", self.dead)]

    def close(self):
        self.q.put(None)
        self.thread.join(timeout=5)
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=30)
        except Exception:
            self.proc.kill()


class ImageWriter:
    """``save(image, path, **save_kwargs)`` like ``image.save(path, **save_kwargs)``, asynchronously.  ``workers=0`` writes
    inline (the reference's behaviour)."""

    def __init__(self, workers: int = 4, depth: int = 8):
        self.n, self.depth = max(0, int(workers)), depth
        self._w: list[_Worker] = []
        self._next = 0
        atexit.register(self.close)

    def save(self, image, path: str, **kw) -> None:
        if self.n == 0 or image.mode not in ("RGB", "L", "RGBA"):
            image.save(path, **kw)
            return
        if not self._w:
            self._w = [_Worker(self.depth) for _ in range(self.n)]
        if "icc_profile" not in kw and image.info.get("icc_profile"):
            kw["icc_profile"] = image.info["icc_profile"]        # Image.save reads it from .info; frombytes() has none
        w = self._w[self._next % self.n]
        self._next += 1
        w.q.put(("save", path, image.mode, image.size, image.tobytes(), kw))

    def flush(self) -> list:
        """wait for every queued file; -> [(path, error text)] of the ones that failed"""
        errs = []
        for w in self._w:
            errs += w.sync()
        return errs

    def close(self) -> None:
        ws, self._w = self._w, []
        for w in ws:
            w.close()


# ---------------------------------------------------------------------------------------------------------------------
# corpus decoding for stage 1: JPEG decode + CLIP's Resize(224, BICUBIC) + CenterCrop(224) in worker processes.
# The reference does this on one CPU thread, one image per model call (retrieval/…:270-287); a thread pool scales poorly
# (the main thread still pays one upload + resize launch per image).  Workers return the 224x224x3 uint8 crop (150 KB), the
# parent only stacks batches.  The crop is PIL's own BICUBIC — the bytes the GPU resample kernel reproduces — so the
# embeddings do not depend on which route an image took.
# ---------------------------------------------------------------------------------------------------------------------
_DECODER = r"""
import pickle, sys
from PIL import Image
inp, out = sys.stdin.buffer, sys.stdout.buffer
size = int(sys.argv[1])
while True:
    try:
        idx, path = pickle.load(inp)
    except EOFError:
        break
    try:
        img = Image.open(path).convert("RGB")
        w, h = img.size
        nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)       # torchvision Resize(int) truncates
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
        msg = (idx, True, img.crop((left, top, left + size, top + size)).tobytes())
    except Exception as e:
        msg = (idx, False, str(e))
    pickle.dump(msg, out, protocol=pickle.HIGHEST_PROTOCOL); out.flush()
"""


class ClipDecodePool:
    """``for idx, ok, payload in pool.run(paths)``: payload = bytes of the uint8 [size,size,3] crop, or the error text.
    Results come back in path order; ``ahead`` bounds the images in flight per worker."""

    def __init__(self, workers: int, size: int = 224, ahead: int = 8):
        self.size, self.ahead = size, ahead
        self.procs = [subprocess.Popen([sys.executable, "-c", _DECODER, str(size)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
                      for _ in range(max(1, workers))]

    def run(self, paths):
        n, W = len(paths), len(self.procs)
        # image k goes to worker k % W and each worker answers in order, so the consumer reads the W bounded result queues
        # round-robin; back-pressure is the queue bound -> the reader stops reading -> the worker blocks on its stdout pipe
        outq = [queue.Queue(maxsize=self.ahead) for _ in range(W)]

        def feed(wi):                                    # blocking pipe writes (tiny messages)
            p = self.procs[wi]
            try:
                for k in range(wi, n, W):
                    pickle.dump((k, paths[k]), p.stdin, protocol=pickle.HIGHEST_PROTOCOL)
                    p.stdin.flush()
            except Exception:                            # worker gone: its reader reports the remaining images as failed
                return

        def read(wi):
            p = self.procs[wi]
            for k in range(wi, n, W):
                try:
                    msg = pickle.load(p.stdout)
                except Exception as e:                   # worker died: fail its remaining images, keep the run going
                    msg = (k, False, f"decoder process failed: {e!r}")
                outq[wi].put(msg)

        threads = [threading.Thread(target=f, args=(wi,), daemon=True) for wi in range(W) for f in (feed, read)]
        for t in threads:
            t.start()
        for k in range(n):
            msg = outq[k % W].get()
            yield (k, msg[1], msg[2])
        for t in threads:
            t.join(timeout=5)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
                p.wait(timeout=10)
            except Exception:
                p.kill()
        self.procs = []
