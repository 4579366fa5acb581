"""stage-2 retrieval-JSON lookup (hostlogic.top5_similar_images and helpers) against goldens captured from the imported
reference (tests/golden/make_stage2_goldens.py -> stage2_lookup.json): same entries, same path fix-ups, same errors"""
import json
import os
import random

import pytest

from domain_rag_amd import hostlogic as H

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "stage2_lookup.json")))
COCO_DIR = "./retrieval/coco/train2017"


@pytest.fixture()
def tree(tmp_path, monkeypatch):
    for f in G["files"]:
        p = tmp_path / f
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"x")
    monkeypatch.chdir(tmp_path)


def _run(fn, *a):
    try:
        return {"ok": fn(*a)}
    except ValueError:
        return {"raises": "ValueError"}


def test_top5(tree):
    for c in G["top5"]:
        got = _run(H.top5_similar_images, G["trees"][c["tree"]], c["sample"], c["dataset"], c["shot"], COCO_DIR)
        if "raises" in c:
            assert got == {"raises": "ValueError"}, c
        else:
            assert "ok" in got and [[float(a), str(b), int(r)] for a, b, r in got["ok"]] == c["ok"], (c, got)


def test_finders(tree):
    for c in G["find_coco"]:
        assert _run(H.find_coco_sample, G["trees"]["coco"], c["sample"], c["shot"]) == {"ok": c["ok"]}, c
    for c in G["find_neudet"]:
        assert _run(H.find_neudet_sample, G["trees"]["neu"], c["sample"], c["shot"]) == {"ok": c["ok"]}, c


def test_correct_image_path(tree):
    for c in G["correct_path"]:
        assert H.correct_image_path(c["path"], COCO_DIR) == c["ok"], c


def test_random_fallback_is_seedable(tree):
    exp = G["random_fallback"]
    a = H.top5_similar_images(G["trees"]["generic"], "no_such_sample", "ArTaxOr", 5, COCO_DIR, random.Random(3))
    b = H.top5_similar_images(G["trees"]["generic"], "no_such_sample", "ArTaxOr", 5, COCO_DIR, random.Random(3))
    assert a == b and len(a) == exp["n"] and [s for s, _, _ in a] == exp["sims"] and [r for _, _, r in a] == exp["ranks"]
    assert all(os.path.dirname(p) == COCO_DIR for _, p, _ in a) == exp["all_in_coco_dir"]


def test_fuzzed_sample_names(tree):
    """160 sample names assembled from category-like tokens: NEU-DET category parsing / aliases / substring fallbacks, the
    COCO variants and the generic variants, against what the reference returned for the same names"""
    n_fallback = 0
    for c in G["fuzz"]:
        name = c["sample"]
        assert _run(H.find_neudet_sample, G["trees"]["neu"], name, 5) == ({"ok": c["neudet"]["ok"]} if "ok" in c["neudet"] else {"raises": "ValueError"}), c
        assert _run(H.find_coco_sample, G["trees"]["coco"], name, 1) == {"ok": c["coco"]["ok"]}, c
        gt = c["generic_top5"]
        if "skip" in gt:
            continue
        got = _run(H.top5_similar_images, G["trees"]["generic"], name, "ArTaxOr", 5, COCO_DIR, random.Random(1))
        if "random_fallback" in gt:
            n_fallback += 1
            assert "ok" in got and [r for _, _, r in got["ok"]] == [1, 2, 3, 4, 5] and all(os.path.dirname(p) == COCO_DIR for _, p, _ in got["ok"]), c
        elif "raises" in gt:
            assert got == {"raises": "ValueError"}, c
        else:
            assert [[float(a), str(b), int(r)] for a, b, r in got["ok"]] == gt["ok"], (c, got)
