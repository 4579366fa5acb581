"""Generate tests/golden/stage2_lookup.json by IMPORTING the reference's batch_generate_flux_kshot.py (read-only at
/root/reference, diffusers stubbed) and recording what its retrieval-JSON lookup functions return on synthetic JSON
trees and a synthetic file tree.  Run in the build container only:

    python tests/golden/make_stage2_goldens.py

The JSON holds inputs + expected outputs only (no reference source).
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage2_lookup.json")

FILES = ["retrieval/coco/train2017/000000000001.jpg", "retrieval/coco/train2017/000000000002.jpg",
         "retrieval/coco/train2017/000000000003.jpg", "retrieval/coco/train2017/000000000004.png",
         "datasets/extra/a.jpg", "other/b.jpg", "rel/c.jpg"]


def sims(paths, ranks=None, **extra):
    ranks = ranks or list(range(1, len(paths) + 1))
    return [dict(rank=r, similarity=round(1.0 / (1 + 0.1 * r), 6), image_path=p, source_dataset="coco", **extra) for p, r in zip(paths, ranks)]


def trees():
    c = "retrieval/coco/train2017/"
    good = sims([c + "000000000001.jpg", "/abs/nowhere/coco/train2017/000000000002.jpg", "../other/b.jpg",
                 "/DATA_HDD/ly/Flux/retrieval/datasets/extra/a.jpg", "/nonexistent/zzz.jpg", c + "000000000003.jpg", c + "000000000004.png"],
                ranks=[3, 1, 2, 5, 4, 6, 7])
    generic = {"ArTaxOr": {"5_shot": {"img_a": [{"sample_id": "img_a", "image_path": "x", "category": "img_a", "similar_images": good}],
                                      "img-b": [{"sample_id": "img-b", "similar_images": sims([c + "000000000002.jpg"])}],
                                      "UPPER": {"similar_images": sims([c + "000000000003.jpg"])},
                                      "nested": [{"wrap": {"similar_images": sims([c + "000000000001.jpg"])}}],
                                      "nested_list": [{"items": [{"similar_images": []}, {"similar_images": sims([c + "000000000004.png"])}]}],
                                      "empty": [{"similar_images": []}],
                                      "junk": [{"similar_images": ["notadict", {"rank": 1, "similarity": 0.5}, {"rank": 2, "similarity": 0.4, "image_path": ""}]}]},
                           "direct_sample": [{"similar_images": sims([c + "000000000001.jpg"])}],
                           "direct_variant": {"similar_images": sims([c + "000000000002.jpg"])}},
               "NoShot": {"s1": [{"similar_images": sims([c + "000000000003.jpg"])}]}}
    coco = {"coco": {"1_shot": {"000000382438": [{"sample_id": "000000382438", "similar_images": sims([c + "000000000001.jpg"])}],
                                "7": {"similar_images": sims([c + "000000000002.jpg"])},
                                "abc": [{"similar_images": sims([c + "000000000003.jpg"])}]}}}
    neu = {"NEU-DET": {"5_shot": {
        "pitted_surface": [{"sample_id": "pitted_surface_12", "similar_images": sims([c + "000000000001.jpg"])},
                           {"sample_id": "pitted_surface_7", "similar_images": sims([c + "000000000002.jpg"])}],
        "rolled-in_scale": [{"sample_id": "rolled-in_scale_14", "similar_images": sims([c + "000000000003.jpg"])}],
        "crazing": [{"sample_id": "crazing_1", "similar_images": sims([c + "000000000004.png"])},
                    {"sample_id": "crazing_22", "similar_images": sims([c + "000000000001.jpg"])}],
        "inclusion": [{"sample_id": "inclusion_106", "similar_images": []}],
        "patches": [],
        "scratch": [{"sample_id": "scratches_3", "similar_images": sims(["/nonexistent/q.jpg"])}]}}}
    return {"generic": generic, "coco": coco, "neu": neu}


def main():
    class _Any:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Any()
        def __call__(self, *a, **k): return _Any()

    class _AnyModule(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Any

    for name in ("diffusers", "diffusers.utils"):
        m = _AnyModule(name)
        m.__path__ = []
        sys.modules[name] = m
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for f in FILES:
            os.makedirs(os.path.dirname(f), exist_ok=True)
            open(f, "wb").write(b"x")
        spec = importlib.util.spec_from_file_location("ref_stage2", os.path.join(REF, "batch_generate_flux_kshot.py"))
        ref = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(ref)
        ref.DATABASE_TYPE = "coco"
        T = trees()

        def call(fn, *a, **k):
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    return {"ok": fn(*a, **k)}
                except ValueError as e:
                    return {"raises": "ValueError", "msg": str(e)}

        g = {"files": FILES, "trees": T, "top5": [], "find_coco": [], "find_neudet": [], "correct_path": []}
        for tree, ds, shot, names in [
            ("generic", "ArTaxOr", 5, ["img_a", "img_b", "img-b", "upper", "Upper", "nested", "nested_list", "empty", "junk",
                                       "direct_sample", "DIRECT_VARIANT", "direct-variant"]),
            ("generic", "NoShot", 5, ["s1", "S1"]),
            ("coco", "coco", 1, ["000000382438", "382438", "7", "0007", "ABC", "missing"]),
            ("neu", "NEU-DET", 5, ["pitted_surface_12", "pitted_surface_99", "rolled-in_scale_14", "rolled-in_scale_2", "crazing_22",
                                   "crazing_5", "inclusion_106", "patches_1", "scratches_3", "unknown_1", "nounderscore", "pitted_7",
                                   "craze_1", "a-b_c_1"]),
            ("neu", "NEU-DET", 1, ["crazing_1"])]:
            for s in names:
                r = call(ref.get_top5_similar_images_from_json, T[tree], s, ds, shot)
                if "ok" in r:
                    r["ok"] = [[float(a), str(b), int(c)] for a, b, c in r["ok"]]
                g["top5"].append({"tree": tree, "dataset": ds, "shot": shot, "sample": s, **r})
        for s in ["000000382438", "382438", "7", "0007", "ABC", "abc", "missing"]:
            g["find_coco"].append({"sample": s, "shot": 1, **call(ref.find_coco_sample, T["coco"], s, 1)})
        for s in ["pitted_surface_12", "pitted_surface_99", "rolled-in_scale_14", "crazing_22", "crazing_5", "patches_1", "scratches_3",
                  "unknown_1", "nounderscore", "pitted_7", "craze_1", "a-b_c_1", "inclusions_106"]:
            g["find_neudet"].append({"sample": s, "shot": 5, **call(ref.find_neudet_sample, T["neu"], s, 5)})
        for p in ["", "rel/c.jpg", "/x/coco/train2017/000000000001.jpg", "/x/coco/train2017/nofile.jpg", "../other/b.jpg", "./rel/c.jpg",
                  "../missing/b.jpg", "/DATA_HDD/ly/Flux/retrieval/datasets/extra/a.jpg", "/DATA_HDD/ly/Flux/retrieval/datasets/extra/none.jpg",
                  "/somewhere/000000000004.png", "/somewhere/none.png", "/x/miniimagenet/train/n01/img.jpg"]:
            g["correct_path"].append({"path": p, **call(ref.get_correct_image_path, p)})
        # fuzz: sample names assembled from category-like tokens, separators and numbers against the fixed trees
        import random as _r
        rr = _r.Random(7)
        toks = ["pitted", "surface", "pitted_surface", "rolled", "rolled-in", "scale", "rolled-in_scale", "crazing", "craze", "inclusion",
                "inclusions", "patches", "patch", "scratches", "scratch", "x", "ab", "img", "UPPER", "direct", "sample", "variant", "a-b", ""]
        names = set()
        while len(names) < 160:
            k = rr.randint(1, 3)
            parts = [rr.choice(toks) for _ in range(k)]
            name = parts[0]
            for q in parts[1:]:
                name += rr.choice(["_", "-", ""]) + q
            if rr.random() < 0.7:
                name += rr.choice(["_", "-", ""]) + str(rr.choice([1, 3, 7, 12, 14, 22, 99, 106]))
            if rr.random() < 0.15:
                name = name.upper() if rr.random() < 0.5 else name.capitalize()
            names.add(name)
        g["fuzz"] = []
        for name in sorted(names):
            r1 = call(ref.find_neudet_sample, T["neu"], name, 5)
            r2 = call(ref.get_top5_similar_images_from_json, T["generic"], name, "ArTaxOr", 5) if name not in ("",) else {"skip": True}
            if "ok" in r2:
                r2["ok"] = [[float(a), str(b), int(c)] for a, b, c in r2["ok"]]
                if len(r2["ok"]) == 5 and all(os.path.dirname(b) == "./retrieval/coco/train2017" for _, b, _ in r2["ok"]) and \
                        [c for _, _, c in r2["ok"]] == [1, 2, 3, 4, 5] and name not in str(T["generic"]):
                    r2 = {"random_fallback": True}            # unseeded pick: only its shape is comparable
            g["fuzz"].append({"sample": name, "neudet": r1, "generic_top5": r2, "coco": call(ref.find_coco_sample, T["coco"], name, 1)})
        # the random fallback (missing sample, generic dataset): similarities and ranks are fixed, the pick is random
        r = call(ref.get_top5_similar_images_from_json, T["generic"], "no_such_sample", "ArTaxOr", 5)
        g["random_fallback"] = {"n": len(r["ok"]), "sims": [float(a) for a, _, _ in r["ok"]], "ranks": [int(c) for _, _, c in r["ok"]],
                                "all_in_coco_dir": all(os.path.dirname(b) == "./retrieval/coco/train2017" for _, b, _ in r["ok"])}
    finally:
        os.chdir(cwd)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=1, ensure_ascii=False)
    print("wrote", OUT, {k: len(v) if hasattr(v, "__len__") else v for k, v in g.items()})


if __name__ == "__main__":
    main()
