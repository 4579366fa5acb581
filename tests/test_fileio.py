"""The native batch file reader that feeds the GPU JPEG route (csrc/fileio.hip: drag_file_sizes / drag_read_files) — host
code, checkable without a GPU: sizes, contents at the caller's offsets, missing files and directories reported per file."""
import ctypes
import os

import numpy as np


def test_native_reader_sizes_contents_and_errors(built_lib, tmp_path):
    rng = np.random.default_rng(0)
    paths, blobs = [], []
    for i in range(300):
        b = rng.integers(0, 256, int(rng.integers(0, 6000)), dtype=np.uint8).tobytes()
        p = tmp_path / f"{i:04d}.bin"
        p.write_bytes(b)
        paths.append(str(p)); blobs.append(b)
    paths.insert(50, str(tmp_path / "missing.bin")); blobs.insert(50, None)
    paths.insert(60, str(tmp_path)); blobs.insert(60, None)                 # a directory is not a file
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    sizes = np.empty(n, np.int64)
    assert built_lib.drag_file_sizes(arr, n, sizes.ctypes.data, 8) == 0
    assert sizes[50] == -2 and sizes[60] < 0                                # -ENOENT, -EISDIR
    assert [int(s) for i, s in enumerate(sizes) if blobs[i] is not None] == [len(b) for b in blobs if b is not None]
    off = np.zeros(n + 1, np.int64)
    np.cumsum(np.maximum(sizes, 0), out=off[1:])
    buf = np.full(int(off[-1]) + 64, 0xAB, np.uint8)
    status = np.full(n, 77, np.int32)
    assert built_lib.drag_read_files(arr, n, buf.ctypes.data, off.ctypes.data, status.ctypes.data, 8) == 0
    assert (status == 0).all()                                              # empty slots are skipped, not errors
    for i, b in enumerate(blobs):
        if b is not None:
            assert buf[off[i]: off[i + 1]].tobytes() == b
    assert (buf[off[-1]:] == 0xAB).all()                                    # nothing written past the last slot
    # a file that shrank after it was sized: reported, the rest unaffected
    open(paths[3], "wb").write(b"xy")
    status[:] = 0
    assert built_lib.drag_read_files(arr, n, buf.ctypes.data, off.ctypes.data, status.ctypes.data, 4) == 0
    assert status[3] == -1 and (np.delete(status, 3) == 0).all()
    assert built_lib.drag_file_sizes(None, 1, sizes.ctypes.data, 1) != 0 and b"bad arguments" in built_lib.drag_last_error()
