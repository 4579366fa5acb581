"""Generate tests/golden/host_logic.json by IMPORTING the reference's own Python (read-only at
/root/reference) with its un-installable dependencies stubbed, and recording what its host-logic functions
return on fixed inputs.  Run in the build container only (the reference never travels):

    python tests/golden/make_host_goldens.py

The JSON holds inputs + expected outputs only (no reference source).
"""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_logic.json")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(path, modname):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import numpy as np
    import torch
    from PIL import Image

    class _Any:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Any()
        def __call__(self, *a, **k): return _Any()

    class _AnyModule(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Any

    for name in ("diffusers", "diffusers.utils", "diffusers.pipelines", "diffusers.pipelines.stable_diffusion"):
        m = _AnyModule(name)
        m.__path__ = []
        sys.modules[name] = m
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        s3 = _load(os.path.join(REF, "outpainting_updown_sampling_redux.py"), "ref_stage3")
        _stub("clip", load=_Any)
        _stub("faiss", IndexFlatIP=_Any)
        _stub("cv2")
        tv = _stub("torchvision"); tvm = _stub("torchvision.models", resnet50=_Any); tv.models = tvm
        os.makedirs(os.path.join(tmp, "retrieval"), exist_ok=True)
        os.chdir(os.path.join(tmp, "retrieval"))
        env_before = os.environ.get("CUDA_VISIBLE_DEVICES")
        s1 = _load(os.path.join(REF, "retrieval", "clip100_resnet_style_all_shots.py"), "ref_stage1")
        if env_before is None:
            os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    finally:
        os.chdir(cwd)

    g = {}
    # ---- resolution policy
    cases = []
    sizes = [(500, 375), (4000, 3000), (1024, 1024), (1023, 2000), (640, 480), (2800, 2800), (2801, 1500), (300, 5000),
             (1365, 1024), (100, 100), (1024, 3000), (2048, 1000), (3000, 1024), (333, 777), (1920, 1080), (1080, 1920)]
    for (w, h) in sizes:
        for mind in (1024, 2048):
            img = Image.new("RGB", (w, h))
            try:
                out, up, down, wu, wd = s3.process_image_resolution(img, mind, 2800)
                cases.append({"w": w, "h": h, "min": mind, "out": list(out.size), "up": up, "down": down, "wu": wu, "wd": wd})
            except ValueError:
                cases.append({"w": w, "h": h, "min": mind, "error": True})
    g["process_image_resolution"] = cases
    g["downscale_image"] = [{"w": w, "h": h, "s": s, "out": list(s3.downscale_image(Image.new("RGB", (w, h)), s).size)}
                            for (w, h, s) in [(1365, 1024, 2.7306666), (1024, 1024, 1.0), (2048, 1536, 3.2), (1707, 1024, 1.6)]]
    # ---- masks
    masks = []
    for (W, H, bbs) in [(64, 48, [[10, 10, 20, 15]]), (64, 48, [[0, 0, 64, 48]]), (50, 40, [[-5, -3, 20, 10], [30, 30, 40, 40]]),
                        (33, 21, [[32, 20, 5, 5]]), (40, 40, [[5.5, 6.5, 10.2, 3.7]]), (16, 16, []),
                        (1365, 1024, [[273, 546, 819, 204], [10, 20, 30, 40]])]:
        m, _ = s3.generate_outpaint_mask(Image.new("RGB", (W, H)), bbs)
        a = np.asarray(m)
        masks.append({"W": W, "H": H, "bboxes": bbs, "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                      "white": int((a == 255).sum()), "mode": m.mode})
    g["generate_outpaint_mask"] = masks
    # ---- randomised sweeps (seeded): more of the same two functions, arbitrary sizes / float boxes / boxes off the canvas
    rs = np.random.default_rng(11)
    sweep = []
    for _ in range(300):
        w, h = int(rs.integers(16, 5000)), int(rs.integers(16, 5000))
        mind = int(rs.choice([512, 1024, 2048]))
        try:
            out, up, down, wu, wd = s3.process_image_resolution(Image.new("RGB", (w, h)), mind, 2800)
            sweep.append([w, h, mind, out.size[0], out.size[1], up, down, wu, wd])
        except ValueError:
            sweep.append([w, h, mind, None])
    g["process_image_resolution_sweep"] = sweep
    msweep = []
    for _ in range(120):
        W, H = int(rs.integers(8, 400)), int(rs.integers(8, 400))
        bbs = []
        for _ in range(int(rs.integers(0, 4))):
            x, y = float(rs.uniform(-0.3 * W, 1.2 * W)), float(rs.uniform(-0.3 * H, 1.2 * H))
            bw, bh = float(rs.uniform(-5, 0.8 * W)), float(rs.uniform(-5, 0.8 * H))
            if rs.random() < 0.5:
                x, y, bw, bh = int(x), int(y), int(bw), int(bh)
            bbs.append([x, y, bw, bh])
        try:
            m, _ = s3.generate_outpaint_mask(Image.new("RGB", (W, H)), bbs)
            a = np.asarray(m)
            msweep.append({"W": W, "H": H, "bboxes": bbs, "sha256": hashlib.sha256(a.tobytes()).hexdigest(), "white": int((a == 255).sum())})
        except Exception as e:
            msweep.append({"W": W, "H": H, "bboxes": bbs, "error": type(e).__name__})
    g["generate_outpaint_mask_sweep"] = msweep
    # ---- sharding / merge
    g["split_samples_for_gpus"] = [{"n": n, "g": k, "sizes": [len(c) for c in s3.split_samples_for_gpus(list(range(n)), k)],
                                    "first": [c[0] if c else None for c in s3.split_samples_for_gpus(list(range(n)), k)]}
                                   for n in range(0, 41) for k in range(1, 9)]
    s3.PROCESS_ID = "golden"
    logs = [{"status": "completed", "sample_id": "a", "category": "cat", "sample_prefix": "DS_a_1shot", "image_id": 3,
             "original_image_size": [10, 20], "bbox_coords_list": [[1, 2, 3, 4]],
             "outpainted_images": [{"original_bg_path": "o", "copied_bg_path": "c", "hires_result_path": "h",
                                    "final_result_path": "f", "mask_path": "m", "params_path": "p", "params": {"k": 1}}]},
            {"status": "error", "sample_id": "b", "outpainted_images": []},
            {"status": "completed", "sample_id": "c", "sample_prefix": "x", "image_id": 1, "original_image_size": [1, 1],
             "outpainted_images": []}]
    fj = s3.generate_formatted_result_json("DS", logs, 5)
    fj.pop("timestamp")
    g["generate_formatted_result_json"] = {"logs": logs, "shot": 5, "out": fj}
    r0 = {"dataset": "DS", "samples": [1, 2], "gpu_process_id": "p_gpu0", "process_id": "p"}
    r1 = {"dataset": "DS", "samples": [3], "gpu_process_id": "p_gpu1"}
    g["merge_gpu_results"] = {"in": [r0, r1], "out": s3.merge_gpu_results("DS", [r0, r1], 1)}
    g["tables"] = {"strength": s3.strength_params, "guidance": s3.guidance_scale_params,
                   "image_prompt_scale": s3.image_prompt_scale_params, "upscale": s3.upscale_dimension_params,
                   "redux_prompt": s3.redux_prompt_params, "default_strength": s3.default_strength,
                   "default_guidance": s3.default_guidance_scale, "min_dim": s3.MIN_DIMENSION, "max_dim": s3.MAX_DIMENSION}
    g["create_gpu_process_id"] = s3.create_gpu_process_id("7", 3)
    # ---- annotation lookup + bbox crops (get_bbox_and_original_image), on a synthetic COCO-format dataset
    ann = {"images": [{"id": 7, "file_name": "abc_001.jpg"}, {"id": "8", "file_name": "zzz.png"}, {"id": 9, "file_name": "noann.jpg"},
                      {"id": 10, "file_name": "missingfile.jpg"}, {"id": 11, "file_name": "sub/dir_img.jpg"}],
           "annotations": [{"image_id": "7", "bbox": [1, 2, 3, 4], "category_id": 2}, {"image_id": 7, "bbox": [5.9, 6.2, 700, 8], "category_id": 9},
                           {"image_id": 8, "bbox": ["-3", "5.9", "1000", "2"], "category_id": 2}, {"image_id": 10, "bbox": [0, 0, 1, 1], "category_id": 2},
                           {"image_id": 11, "bbox": [63, 47, 0, 0], "category_id": 3}],
           "categories": [{"id": 2, "name": "beetle"}, {"id": 3, "name": "moth"}]}
    dtmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(dtmp, "DS", "annotations")); os.makedirs(os.path.join(dtmp, "DS", "train", "sub"))
    json.dump(ann, open(os.path.join(dtmp, "DS", "annotations", "1_shot.json"), "w"))
    sizes_on_disk = {"abc_001.jpg": (64, 48), "zzz.png": (100, 50), "noann.jpg": (8, 8), "sub/dir_img.jpg": (64, 48)}
    for fn, sz in sizes_on_disk.items():
        Image.new("RGB", sz).save(os.path.join(dtmp, "DS", "train", fn))
    s3.datasets_dir = dtmp
    s3.load_image = lambda path: Image.open(path).convert("RGB")     # diffusers.utils.load_image on a local file (stubbed module)
    cases = []
    for sid in ["abc_001", "abc", "xx_abc_001_yy", "zzz", "noann", "missingfile", "nope", "dir_img", "sub/dir_img"]:
        img, crops, boxes, image_id, cats = s3.get_bbox_and_original_image("DS", sid, 1)
        cases.append({"sample_id": sid, "found": img is not None, "image_size": list(img.size) if img is not None else None,
                      "crop_sizes": [list(c.size) if c is not None else None for c in crops] if crops is not None else None,
                      "bboxes": boxes, "image_id": image_id, "categories": cats})
    g["get_bbox_and_original_image"] = {"annotations": ann, "files": {k: list(v) for k, v in sizes_on_disk.items()}, "cases": cases}
    # ---- stage 1 helpers
    paths = ["../../pipeline/datasets/coco/train2017/1.jpg", "../../datasets/coco/val2017/2.jpg", "./coco/3.jpg", 5, None,
             "/abs/../../datasets/coco/x.jpg"]
    g["clean_image_path"] = [{"in": p, "out": s1.clean_image_path(p)} for p in paths]
    # ---- two-stage retrieval plumbing around the (stubbed) numeric kernels
    # second stage: style vectors are injected, so this pins distance / stable sort / similarity / rank / path cleaning
    rng = np.random.default_rng(5)
    style = {f"../../datasets/coco/train2017/{i}.jpg": rng.standard_normal(128).astype(np.float32) for i in range(12)}
    style["../../datasets/coco/train2017/3.jpg"] = style["../../datasets/coco/train2017/7.jpg"].copy()      # a tie
    style["../../datasets/coco/train2017/5.jpg"] = None                                                      # unreadable
    style["q.jpg"] = rng.standard_normal(128).astype(np.float32)
    cleaned = {s1.clean_image_path(k): v for k, v in style.items()}
    s1.compute_resnet_features = lambda path, model, device: cleaned.get(path)
    first = [{"similarity": float(1.0 - 0.01 * i), "image_path": f"../../datasets/coco/train2017/{i}.jpg", "index": i,
              **({"source_dataset": "coco"} if i % 4 else {})} for i in range(12)]
    g["resnet_second_stage_rerank"] = {
        "style": {k: (v.tolist() if v is not None else None) for k, v in cleaned.items()}, "query": "q.jpg", "first": first,
        "out": s1.resnet_second_stage_rerank("q.jpg", first, None, None),
        "out_query_unreadable": s1.resnet_second_stage_rerank("unreadable.jpg", first[:3], None, None)}
    # first stage: exact inner product with a numpy stand-in for faiss (scores are well separated: order is unambiguous)
    class _NumpyFlatIP:
        def __init__(self, d): self.x = np.zeros((0, d), np.float32)
        def add(self, x): self.x = np.vstack([self.x, x])
        def search(self, q, k):
            sc = q @ self.x.T
            I = np.argsort(-sc, axis=1, kind="stable")[:, :k]
            return np.take_along_axis(sc, I, 1), I.astype(np.int64)
    s1.faiss.IndexFlatIP = _NumpyFlatIP
    q = np.zeros(64, np.float32); q[0] = 1.0
    fa = np.zeros((5, 64), np.float32); fa[:, 0] = [0.9, 0.1, 0.5, 0.7, 0.3]
    fb = np.zeros((3, 64), np.float32); fb[:, 0] = [0.8, 0.2, 0.6]
    feats = {"coco": fa, "miniimagenet": fb, "empty": np.zeros((0, 64), np.float32), "none": None}
    pths = {"coco": [f"c{i}.jpg" for i in range(5)], "miniimagenet": [f"m{i}.jpg" for i in range(3)], "empty": [], "none": None}
    g["clip_first_stage_retrieval"] = {"query": q.tolist(), "features": {k: (v.tolist() if v is not None else None) for k, v in feats.items()},
                                       "paths": pths, "out_k100": s1.clip_first_stage_retrieval(q, feats, pths, top_k=100),
                                       "out_k3": s1.clip_first_stage_retrieval(q, feats, pths, top_k=3),
                                       "out_nothing": s1.clip_first_stage_retrieval(q, {"none": None}, {"none": None}, top_k=3)}
    m, s = s1.calc_mean_std(torch.arange(96.0).reshape(2, 3, 4, 4))
    g["calc_mean_std"] = {"mean": m.flatten().tolist(), "std": s.flatten().tolist()}
    with open(OUT, "w") as f:
        json.dump(g, f, indent=1, ensure_ascii=False)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
