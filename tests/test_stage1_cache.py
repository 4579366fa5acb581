"""stage-1 feature-cache resolution (which file wins, key fallbacks, path cleaning, when to recompute) against goldens
captured from the imported reference (tests/golden/make_stage1_goldens.py -> stage1_cache.json)"""
import importlib.util
import json
import os

import numpy as np

from domain_rag_amd.cli import stage1_retrieval as S1

HERE = os.path.dirname(__file__)
G = json.load(open(os.path.join(HERE, "golden", "stage1_cache.json")))


def test_cache_resolution_matches_reference(tmp_path):
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_stage1_goldens.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)                      # only its write_case_files() helper is used (no reference import)
    mk.write_case_files(str(tmp_path))
    for c in G["cases"]:
        rd = tmp_path / ("no_results" if c.get("no_local") else "results")
        rd.mkdir(exist_ok=True)
        hit = S1.resolve_feature_cache(str(tmp_path / c["feats"]) if c["feats"] else None, str(tmp_path / c["paths"]) if c["paths"] else None,
                                       str(rd / "coco_clip_features.npy"), str(rd / "coco_image_paths.json"), bool(c.get("force")))
        if c["result"] is None:
            assert hit is None, c
        else:
            assert hit is not None, c
            f, p = hit
            assert len(f) == c["result"]["n"] and list(p) == c["result"]["paths"] and abs(float(np.asarray(f).reshape(len(f), -1)[0, 1]) - c["result"]["first"]) < 1e-9, c
