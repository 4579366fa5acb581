"""Generate tests/golden/stage1_cache.json by IMPORTING the reference's retrieval script (read-only at /root/reference,
clip / faiss / cv2 / torchvision stubbed) and recording which cache source its ``load_or_compute_coco_features`` picks and
what it returns for crafted cache files.  Run in the build container only:

    python tests/golden/make_stage1_goldens.py

The JSON holds inputs + expected outputs only (no reference source).
"""
import argparse
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_cache.json")
PATHS_A = ["../../pipeline/datasets/coco/train2017/1.jpg", "./coco/2.jpg", "../../datasets/coco/val2017/3.jpg"]
PATHS_B = ["b0.jpg", "../../datasets/coco/train2017/b1.jpg"]


def write_case_files(d):
    """the cache files every case may refer to (same function is used by the test to recreate them)"""
    fa = np.arange(3 * 512, dtype=np.float32).reshape(3, 512) / 1000.0
    fb = -np.arange(2 * 512, dtype=np.float32).reshape(2, 512) / 1000.0
    torch.save({"embeddings": torch.from_numpy(fa), "image_paths": PATHS_A}, os.path.join(d, "emb_paths.pt"))
    torch.save({"features": torch.from_numpy(fa), "paths": PATHS_A}, os.path.join(d, "feat_paths.pt"))
    torch.save({"features": torch.from_numpy(fa)}, os.path.join(d, "feat_only.pt"))
    torch.save({"something": torch.from_numpy(fa)}, os.path.join(d, "odd_keys.pt"))
    torch.save(torch.from_numpy(fa), os.path.join(d, "raw_tensor.pt"))
    np.save(os.path.join(d, "feats.npy"), fa)
    json.dump(PATHS_A, open(os.path.join(d, "paths_a.json"), "w"))
    json.dump(PATHS_B, open(os.path.join(d, "paths_b.json"), "w"))
    os.makedirs(os.path.join(d, "results"), exist_ok=True)
    np.save(os.path.join(d, "results", "coco_clip_features.npy"), fb)
    json.dump(PATHS_B, open(os.path.join(d, "results", "coco_image_paths.json"), "w"))


CASES = [
    dict(name="pt embeddings + image_paths", feats="emb_paths.pt", paths=None),
    dict(name="pt features + paths", feats="feat_paths.pt", paths=None),
    dict(name="pt features only + json", feats="feat_only.pt", paths="paths_a.json"),
    dict(name="pt features only, no json", feats="feat_only.pt", paths=None),
    dict(name="pt raw tensor + json", feats="raw_tensor.pt", paths="paths_a.json"),
    dict(name="pt in-file paths AND json (json wins)", feats="emb_paths.pt", paths="paths_b.json"),
    dict(name="npy + json", feats="feats.npy", paths="paths_a.json"),
    dict(name="npy without json", feats="feats.npy", paths=None),
    dict(name="missing file -> local cache", feats="nope.pt", paths=None),
    dict(name="nothing given -> local cache", feats=None, paths=None),
    dict(name="force recompute", feats="emb_paths.pt", paths=None, force=True),
    dict(name="nothing given, no local cache", feats=None, paths=None, no_local=True),
]


def main():
    class _Any:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Any()
        def __call__(self, *a, **k): return _Any()
    for name, attrs in (("clip", dict(load=_Any)), ("faiss", dict(IndexFlatIP=_Any)), ("cv2", {})):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tvm.resnet50 = _Any; tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.makedirs(os.path.join(tmp, "retrieval"))
    os.chdir(os.path.join(tmp, "retrieval"))
    env_before = os.environ.get("CUDA_VISIBLE_DEVICES")
    try:
        spec = importlib.util.spec_from_file_location("ref_stage1", os.path.join(REF, "retrieval", "clip100_resnet_style_all_shots.py"))
        s1 = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(s1)
        if env_before is None:
            os.environ.pop("CUDA_VISIBLE_DEVICES", None)
        write_case_files(tmp)
        s1.compute_coco_clip_features = lambda *a, **k: (np.zeros((0, 512), np.float32), [])      # "fell through to recompute"
        out = []
        for c in CASES:
            s1.RESULTS_DIR = os.path.join(tmp, "no_results" if c.get("no_local") else "results")
            os.makedirs(s1.RESULTS_DIR, exist_ok=True)
            args = argparse.Namespace(force_recompute=bool(c.get("force")), global_features=False, coco_dir="./coco",
                                      pretrained_coco_features=os.path.join(tmp, c["feats"]) if c["feats"] else None,
                                      pretrained_coco_paths=os.path.join(tmp, c["paths"]) if c["paths"] else None)
            with contextlib.redirect_stdout(io.StringIO()):
                f, p = s1.load_or_compute_coco_features(args, "cpu", None, None)
            if f is None:
                out.append({**c, "result": None})
            else:
                arr = f.detach().cpu().numpy() if torch.is_tensor(f) else np.asarray(f)
                out.append({**c, "result": {"n": int(len(arr)), "first": float(arr.reshape(len(arr), -1)[0, 1]) if len(arr) else None, "paths": list(p)}})
    finally:
        os.chdir(cwd)
    json.dump({"paths_a": PATHS_A, "paths_b": PATHS_B, "cases": out}, open(OUT, "w"), indent=1, ensure_ascii=False)
    for c in out:
        print(c["name"], "->", c["result"] if c["result"] is None else (c["result"]["n"], c["result"]["first"], c["result"]["paths"][:1]))


if __name__ == "__main__":
    main()
