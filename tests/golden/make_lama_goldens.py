"""Generate tests/golden/lama_stage.json by IMPORTING the reference's LaMa script (read-only at /root/reference,
``simple_lama_inpainting`` replaced by a recording stand-in) and capturing
  * ``create_mask_from_multiple_bboxes`` / ``create_mask_from_bbox`` on crafted and random boxes (ints, floats, negative and
    out-of-frame corners, degenerate boxes) — the masks as run-length rows;
  * what ``process_dataset`` hands to the model and writes for a crafted mini dataset: call order, image / mask sizes and
    checksums, output paths (incl. the ``-`` -> ``_`` directory rename), the (processed, errors) counters.
Run in the build container only:

    python tests/golden/make_lama_goldens.py

The JSON holds inputs + expected outputs only (no reference source).
"""
import contextlib
import hashlib
import importlib.util
import io
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lama_stage.json")

MASK_CASES = [
    (64, 48, [[10, 10, 20, 15]]),
    (64, 48, [[10.7, 9.2, 20.5, 15.9]]),
    (64, 48, [[-5, -3, 20, 15]]),                    # x is clamped FIRST, the right edge is computed from the clamped x
    (64, 48, [[50, 40, 30, 30]]),                    # right / bottom clamp to the frame size (one past the last pixel)
    (64, 48, [[63, 47, 1, 1]]),
    (64, 48, [[64, 10, 5, 5]]),                      # starts outside: right == x -> skipped
    (64, 48, [[10, 10, 0, 5]]),                      # zero width -> skipped
    (64, 48, [[10, 10, 0.4, 0.4]]),                  # sub-pixel box still draws the pixel it starts in
    (64, 48, [[10, 10, -3, 5]]),
    (64, 48, [[1, 2, 3, 4], [30.5, 20.5, 10.25, 10.75], [60, 44, 10, 10]]),
    (17, 9, [[0, 0, 17, 9]]),
    (17, 9, [[16.9, 8.9, 5, 5]]),
    (33, 21, []),
]


def rle_rows(mask):
    """[[row, start, stop_exclusive], ...] of the non-zero runs (all values are 0 / 255)"""
    runs = []
    for y, row in enumerate(np.asarray(mask)):
        nz = np.flatnonzero(row)
        if nz.size:
            splits = np.flatnonzero(np.diff(nz) > 1)
            starts = np.concatenate([[nz[0]], nz[splits + 1]])
            stops = np.concatenate([nz[splits], [nz[-1]]]) + 1
            runs += [[int(y), int(a), int(b)] for a, b in zip(starts, stops)]
    return runs


def load_reference(record):
    stub = types.ModuleType("simple_lama_inpainting")

    class SimpleLama:
        def __init__(self):
            record.append(["init"])

        def __call__(self, image, mask):
            a, m = np.asarray(image), np.asarray(mask)
            record.append(["call", image.mode, list(image.size), mask.mode, list(mask.size), hashlib.sha1(a.tobytes()).hexdigest(),
                           rle_rows(m), sorted(int(v) for v in np.unique(m))])
            if a[0, 0, 0] == 7:                       # crafted failure: the reference counts it as an error and goes on
                raise ValueError("boom")
            return Image.fromarray(255 - a)

    stub.SimpleLama = SimpleLama
    sys.modules["simple_lama_inpainting"] = stub
    spec = importlib.util.spec_from_file_location("ref_lama_inpaint", os.path.join(REF, "lama_inpaint", "lama_inpaint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write_dataset(root):
    """the crafted mini dataset (the test recreates it with this same function): cwd = <root>/lama_inpaint"""
    rng = np.random.default_rng(5)
    ds = os.path.join(root, "datasets", "NEU-DET")
    os.makedirs(os.path.join(ds, "train", "sub"), exist_ok=True)
    os.makedirs(os.path.join(ds, "annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "lama_inpaint"), exist_ok=True)
    images, anns = [], []

    def add(iid, name, arr, w=None, h=None, mode=None):
        im = Image.fromarray(arr)
        if mode:
            im = im.convert(mode)
        im.save(os.path.join(ds, "train", name))
        images.append({"id": iid, "file_name": name, "width": w or arr.shape[1], "height": h or arr.shape[0]})

    add(1, "a.png", rng.integers(0, 256, (40, 56, 3), dtype=np.uint8))
    add(2, "b_gray.png", rng.integers(0, 256, (33, 47), dtype=np.uint8))                       # L mode -> convert("RGB")
    add(3, "sub/c.png", rng.integers(0, 256, (30, 30, 3), dtype=np.uint8), w=45, h=36)           # annotation size differs -> resize
    fail = rng.integers(0, 256, (24, 24, 3), dtype=np.uint8); fail[0, 0, 0] = 7
    add(4, "d_fail.png", fail)
    add(5, "e_rgba.png", rng.integers(0, 256, (20, 28, 4), dtype=np.uint8))                       # RGBA -> convert("RGB")
    add(6, "f_unannotated.png", rng.integers(0, 256, (16, 16, 3), dtype=np.uint8))               # no annotation: never processed
    anns += [{"id": 1, "image_id": 1, "bbox": [5, 6, 20, 10], "category_id": 1},
             {"id": 2, "image_id": 1, "bbox": [30.5, 20.25, 40, 40], "category_id": 2},
             {"id": 3, "image_id": 2, "bbox": [0, 0, 10, 10], "category_id": 1},
             {"id": 4, "image_id": 3, "bbox": [10, 8, 12.5, 9.5], "category_id": 9},
             {"id": 5, "image_id": 4, "bbox": [2, 2, 5, 5], "category_id": 1},
             {"id": 6, "image_id": 5, "bbox": [3, 3, 8, 8], "category_id": 2},
             {"id": 7, "image_id": 77, "bbox": [1, 1, 2, 2], "category_id": 1},                 # image id without an image entry
             {"id": 8, "image_id": 8, "bbox": [1, 1, 2, 2], "category_id": 1}]
    images.append({"id": 8, "file_name": "missing.png", "width": 10, "height": 10})              # file does not exist -> error
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "crazing"}, {"id": 2, "name": "patches"}]},
              open(os.path.join(ds, "annotations", "1_shot.json"), "w"))


def main():
    record = []
    ref = load_reference(record)
    gold = {"masks": [], "random_masks": [], "dataset": {}}
    for w, h, boxes in MASK_CASES:
        m = ref.create_mask_from_multiple_bboxes(w, h, boxes)
        entry = {"w": w, "h": h, "boxes": boxes, "runs": rle_rows(m), "values": sorted(int(v) for v in np.unique(np.asarray(m)))}
        if len(boxes) == 1:
            assert rle_rows(ref.create_mask_from_bbox(w, h, boxes[0])) == entry["runs"]
        gold["masks"].append(entry)
    rng = np.random.default_rng(11)
    for _ in range(60):
        w, h = int(rng.integers(8, 90)), int(rng.integers(8, 90))
        boxes = []
        for _ in range(int(rng.integers(1, 4))):
            b = rng.uniform(-10, 80, 4)
            b[2:] = rng.uniform(-2, 50, 2)
            boxes.append([float(round(v, 2)) if rng.random() < 0.5 else int(v) for v in b])
        m = ref.create_mask_from_multiple_bboxes(w, h, boxes)
        gold["random_masks"].append({"w": w, "h": h, "boxes": boxes, "sha1": hashlib.sha1(np.asarray(m).tobytes()).hexdigest()})
    with tempfile.TemporaryDirectory() as root:
        write_dataset(root)
        cwd = os.getcwd()
        os.chdir(os.path.join(root, "lama_inpaint"))
        try:
            logger = logging.getLogger("golden_lama")
            logger.addHandler(logging.NullHandler()); logger.propagate = False
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                counts = ref.process_dataset("NEU-DET", "1", logger)
                missing = ref.process_dataset("nope", "1", logger)
        finally:
            os.chdir(cwd)
        outs = {}
        base = os.path.join(root, "lamainpaint")
        for dp, _, fs in os.walk(base):
            for f in fs:
                full = os.path.join(dp, f)
                outs[os.path.relpath(full, base)] = [list(Image.open(full).size), Image.open(full).mode,
                                                     hashlib.sha1(np.asarray(Image.open(full)).tobytes()).hexdigest()]
        gold["dataset"] = {"counts": list(counts), "missing_dataset_counts": list(missing), "calls": record, "outputs": outs}
    json.dump(gold, open(OUT, "w"), indent=1)
    print(f"wrote {OUT}: {len(gold['masks'])} masks, {len(gold['random_masks'])} random, {len(record)} model events, outputs {sorted(outs)}")


if __name__ == "__main__":
    main()
