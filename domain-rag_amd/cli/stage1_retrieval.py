"""Stage 1 — drop-in for ``retrieval/clip100_resnet_style_all_shots.py`` (flags :966-998, outputs :866-895,:1088-1097).

    python -m domain_rag_amd.cli.stage1_retrieval --datasets ArTaxOr --shots 1 5 [--coco-dir ./coco] ...

Same flags, cache files and JSON schemas; differences are the deliberate ones of SURVEY §9 (the environment's
CUDA_VISIBLE_DEVICES is honoured, --clip-top-k is live, the index is built once, corpus embedding is batched and — under
torch.distributed.run — sharded across ranks with one all-gather).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
from pathlib import Path

import numpy as np
import torch

from .. import hostlogic as H, retrieval as R

RESULTS_DIR = "./retrieval_results"
LAMAINPAINT_DIR = "../lamainpaint"


CLIP_DEFAULT = os.path.join("~", ".cache", "clip", "ViT-B-32.pt")
RESNET_DEFAULT = os.path.join("~", ".cache", "torch", "hub", "checkpoints", "resnet50-0676ba61.pth")


def resolve_weights(args):
    """(clip_weights_path | None, resnet_weights_path | None).  The reference loads pretrained CLIP ViT-B/32 and torchvision
    IMAGENET1K_V1 weights by default (retrieval/clip100_resnet_style_all_shots.py:54,209); so does this CLI: explicit flag, then
    the environment, then the libraries' own download caches.  Seeded random weights only with --synthetic-weights /
    DRAG_SYNTHETIC_WEIGHTS=1 — never silently (a drop-in run would otherwise write a meaningless retrieval JSON)."""
    synthetic = args.synthetic_weights or os.environ.get("DRAG_SYNTHETIC_WEIGHTS", "") not in ("", "0")
    if synthetic:
        if args.clip_weights or args.resnet_weights:
            raise SystemExit("--synthetic-weights excludes --clip-weights / --resnet-weights")
        return None, None
    out = []
    for flag, given, env, default, what in (("--clip-weights", args.clip_weights, "DRAG_CLIP_WEIGHTS", CLIP_DEFAULT, "CLIP ViT-B/32"),
                                            ("--resnet-weights", args.resnet_weights, "DRAG_RESNET_WEIGHTS", RESNET_DEFAULT,
                                             "torchvision resnet50 (IMAGENET1K_V1)")):
        path = given or os.environ.get(env) or os.path.expanduser(default)
        if not os.path.isfile(path):
            raise SystemExit(f"stage 1: no {what} weights: {path} does not exist.  Pass {flag} <file> (or ${env}); "
                             f"--synthetic-weights runs with seeded random weights instead (tests / benchmarks only).")
        out.append(path)
    return out[0], out[1]


def tensor_fingerprint(tensors) -> str:
    """short sha256 over the raw bytes of a few tensors (identifies a set of weights without hashing hundreds of MB)"""
    import hashlib
    h = hashlib.sha256()
    for t in tensors:
        a = t.detach().to("cpu", torch.float32).contiguous().numpy()
        h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()[:16]


def feature_cache_meta(model) -> dict:
    """what a cached embedding matrix depends on besides the image files"""
    return {"clip": getattr(model, "fingerprint", "unknown"), "precision": getattr(model, "precision", "unknown"),
            "preprocess": "pil-bicubic-224-centercrop"}


def style_cache_meta(stem) -> dict:
    try:
        import cv2  # noqa: F401
        backend = "cv2"
    except ImportError:
        backend = "cv2-restated"
    if getattr(stem, "gpu_files", False) or getattr(stem, "force_restated", False):
        backend = "cv2-restated"            # the device route and its host detour both use the restated tables
    return {"stem": tensor_fingerprint([stem.state[k] for k in sorted(stem.state)]), "resize": backend}


def _meta_path(cache_f: str) -> str:
    return os.path.splitext(cache_f)[0] + ".meta.json"


def local_cache_is_stale(cache_f: str, want: dict) -> bool:
    """A local cache written by THIS CLI carries a side file with the weights / precision it was computed from; when that
    differs from the running model the cache is stale.  A cache without a side file (written by the reference script, or copied
    in) is taken as it is, like the reference takes it."""
    mp = _meta_path(cache_f)
    if not os.path.exists(mp):
        return False
    try:
        with open(mp) as f:
            have = json.load(f)
    except Exception:
        return True
    return any(have.get(k) != v for k, v in want.items())


def build_parser():
    p = argparse.ArgumentParser(description="CLIP+ResNet图像检索 - 多shot版本 (MI355X)")
    p.add_argument("--datasets", type=str, nargs="+", default=["ArTaxOr", "DIOR", "FISH", "NEU-DET", "UODD", "clipart1k"])
    p.add_argument("--shots", type=int, nargs="+", default=[1, 5, 10])
    p.add_argument("--coco-dir", type=str, default="./coco")
    p.add_argument("--mini-imagenet-dir", type=str, default="./miniimagenet")
    p.add_argument("--dataset-source", type=str, choices=["coco", "mini-imagenet", "both"], default="coco")
    p.add_argument("--clip-top-k", type=int, default=100)
    p.add_argument("--output-dir", type=str, default=None)
    p.add_argument("--gpu-id", type=int, default=0)
    p.add_argument("--pretrained-coco-features", type=str, default="./coco_embeddings_global.pt")
    p.add_argument("--pretrained-coco-paths", type=str, default=None)
    p.add_argument("--pretrained-mini-imagenet-features", type=str, default=None)
    p.add_argument("--pretrained-mini-imagenet-paths", type=str, default=None)
    p.add_argument("--global-features", action="store_true")
    p.add_argument("--force-recompute", action="store_true")
    p.add_argument("--lamainpaint-dir", type=str, default=None)
    p.add_argument("--force-recompute-inpainted", action="store_true")
    # additions (not in the reference)
    p.add_argument("--clip-precision", choices=["fp32", "bf16"], default=None,
                   help="arithmetic of the CLIP tower: fp32 (default; what openai-CLIP computes on its CPU path) or the faster bf16 MFMA tower")
    p.add_argument("--clip-weights", type=str, default=None,
                   help="openai CLIP ViT-B/32 checkpoint (ViT-B-32.pt); default: $DRAG_CLIP_WEIGHTS, else openai-CLIP's own download "
                        "cache ~/.cache/clip/ViT-B-32.pt (where the reference's clip.load() leaves it)")
    p.add_argument("--resnet-weights", type=str, default=None,
                   help="torchvision resnet50 state_dict; default: $DRAG_RESNET_WEIGHTS, else torchvision's hub cache "
                        "~/.cache/torch/hub/checkpoints/resnet50-0676ba61.pth (IMAGENET1K_V1, what the reference loads)")
    p.add_argument("--synthetic-weights", action="store_true",
                   help="run with seeded random weights of the same architectures (tests / benchmarks; results are meaningless for retrieval)")
    p.add_argument("--embed-batch", type=int, default=256)
    p.add_argument("--no-visuals", action="store_true", help="skip the per-query *_visual.jpg contact sheets")
    p.add_argument("--decode-procs", type=int, default=-1,
                   help="worker processes for JPEG decode + CLIP resize/crop of the corpus (-1: auto, 0: thread pool + GPU resize)")
    p.add_argument("--decode", choices=["gpu", "host"], default="gpu",
                   help="corpus JPEG decode: on the GPU (default; byte-identical to PIL, the host only reads the files) or on the "
                        "host (PIL in --decode-procs worker processes / threads)")
    p.add_argument("--style-resize", choices=["auto", "cv2", "restated"], default="auto",
                   help="256x256 resize of the ResNet-stem style re-rank (retrieval/clip100_resnet_style_all_shots.py:186-192 calls "
                        "cv2.imread + cv2.resize): 'cv2' = the real OpenCV calls on the host, exactly the reference's pixels; 'restated' = "
                        "OpenCV's INTER_LINEAR restated (parity with cv2 UNPINNED: OpenCV is absent from the build box), which is what "
                        "lets the candidates of a query be decoded + resized on the GPU in one batch; 'auto' (default) = cv2 whenever "
                        "OpenCV is importable, restated otherwise")
    p.add_argument("--host-preprocess", action="store_true",
                   help="resize on the host with PIL like the reference (default: PIL-exact resize on the GPU; same bits)")
    p.add_argument("--style-cache", type=str, default=None,
                   help="npz of corpus style vectors reused across queries and runs (default: <output-dir>/style_cache.npz)")
    return p


def list_corpus_images(root: str, subdirs) -> list[str]:
    """ref :242-258: first existing sub-directory, then **/*.jpg, jpeg, png.  Sorted for a deterministic global row
    order (the reference takes filesystem order; sorting is required for the sharded embedding)."""
    base = None
    for s in subdirs:
        if os.path.isdir(os.path.join(root, s)):
            base = os.path.join(root, s)
            break
    if base is None:
        base = root if os.path.isdir(root) else None
    if base is None:
        return []
    out = []
    for ext in ("jpg", "jpeg", "png"):
        out += sorted(str(p) for p in Path(base).glob(f"**/*.{ext}"))
    return out


def load_feature_file(feat_path, paths_path):
    """the "pretrained features" step of load_or_compute_coco_features (ref :547-606): features from a .pt (dict key
    ``embeddings``, else ``features``, else the object itself) or a .npy; paths from the .pt (``image_paths``, else
    ``paths``), overridden by the json list when one is given.  Returns (features | None, paths list)."""
    feats, paths, data = None, [], None
    if feat_path.endswith(".pt"):
        data = torch.load(feat_path, map_location="cpu", weights_only=False)
        if isinstance(data, dict):
            feats = data["embeddings"] if "embeddings" in data else data["features"] if "features" in data else data
        else:
            feats = data
    elif feat_path.endswith(".npy"):
        feats = np.load(feat_path)
    if isinstance(data, dict) and ("image_paths" in data or "paths" in data):
        paths = [R.clean_image_path(p) for p in (data["image_paths"] if "image_paths" in data else data["paths"])]
    if paths_path and os.path.exists(paths_path):
        with open(paths_path) as f:
            paths = [R.clean_image_path(p) for p in json.load(f)]
    if torch.is_tensor(feats):
        feats = feats.float().cpu().numpy()
    return feats, paths


def resolve_feature_cache(pre_feats, pre_paths, cache_f, cache_p, force_recompute=False, global_candidates=()):
    """cache resolution order of load_or_compute_coco_features (ref :500-622), without the recompute itself:
    global pre-extracted file (if asked) -> the given pretrained file -> the local cache under the results dir.
    Returns (features, paths) or None when the caller has to recompute.  Like the reference, a pretrained file that yields
    features but no paths does NOT fall back to the local cache: it recomputes."""
    if force_recompute:
        return None
    feats = paths = None
    for g in global_candidates:                       # (:511-548) {embeddings|features, image_paths}
        if os.path.exists(g):
            try:
                d = torch.load(g, map_location="cpu", weights_only=False)
                if isinstance(d, dict) and "image_paths" in d and ("embeddings" in d or "features" in d):
                    feats = d["embeddings"] if "embeddings" in d else d["features"]
                    feats = feats.float().cpu().numpy() if torch.is_tensor(feats) else np.asarray(feats)
                    paths = [R.clean_image_path(p) for p in d["image_paths"]]
                    break
            except Exception as e:
                print(f"加载全局特征文件 {g} 时出错: {e}")
    if feats is None and pre_feats is not None:
        if os.path.exists(pre_feats):
            try:
                feats, paths = load_feature_file(pre_feats, pre_paths)
            except Exception as e:
                print(f"加载预提取特征时出错: {e}")
                feats = None
        else:
            print(f"警告：指定的预提取特征文件不存在: {pre_feats}")
    if feats is None and os.path.exists(cache_f) and os.path.exists(cache_p):
        try:
            feats = np.load(cache_f)
            with open(cache_p) as fh:
                paths = [R.clean_image_path(p) for p in json.load(fh)]
        except Exception as e:
            print(f"加载本地缓存特征时出错: {e}")
            feats = None
    if feats is None or paths is None or len(feats) == 0 or len(paths) == 0:
        return None
    return feats, paths


def load_or_compute_features(args, tag, root, subdirs, pre_feats, pre_paths, model, preprocess, results_dir, rank0=True):
    """ref :500-655 cache resolution order, then compute + save ``<tag>_clip_features.npy`` / ``<tag>_image_paths.json``"""
    cache_f = os.path.join(results_dir, f"{tag}_clip_features.npy")
    cache_p = os.path.join(results_dir, f"{tag}_image_paths.json")
    glob_c = (os.path.join("..", f"{tag}_embeddings_global.pt"), os.path.join("..", "result_clip_vision", f"{tag}_embeddings_global.pt")) \
        if getattr(args, "global_features", False) else ()
    meta = feature_cache_meta(model)
    if os.path.exists(cache_f) and local_cache_is_stale(cache_f, meta):
        print(f"本地缓存特征与当前CLIP权重/精度不符，重新计算: {cache_f}")
        cache_f_use, cache_p_use = cache_f + ".stale", cache_p + ".stale"        # names that do not exist: skips the local step
    else:
        cache_f_use, cache_p_use = cache_f, cache_p
    hit = resolve_feature_cache(pre_feats, pre_paths, cache_f_use, cache_p_use, args.force_recompute, glob_c)
    if hit is not None:
        print(f"成功加载 {len(hit[0])} 个预提取特征")
        return np.asarray(hit[0], dtype=np.float32), hit[1]
    paths = list_corpus_images(root, ("images", "train2017", "val2017", "train"))
    if not paths:
        print(f"错误：找不到图像: {root}")
        return None, None
    procs = args.decode_procs
    if procs < 0:                                   # auto: worker processes when the host has cores to spare
        ncpu = os.cpu_count() or 1
        procs = min(32, ncpu // 2) if ncpu >= 8 else 0
    feats, valid = R.compute_corpus_features(model, preprocess, paths, batch=args.embed_batch, decode_procs=procs,
                                             gpu_decode=(args.decode == "gpu" and not args.host_preprocess))
    if rank0 and len(feats):
        np.save(cache_f, feats)
        with open(cache_p, "w") as fh:
            json.dump(valid, fh)
        with open(_meta_path(cache_f), "w") as fh:
            json.dump(meta, fh)
    return feats, valid


def get_inpainted_images(lama_dir, dataset, shot):
    """ref :89-158: ``<lama>/<ds>/<k>_shot/*.jpg``; category = sample id unless category_mapping.json says otherwise.
    Accepts LaMa's ``safe`` spelling (``-`` -> ``_``) as well."""
    shot_dir = os.path.join(lama_dir, dataset, f"{shot}_shot")
    if not os.path.isdir(shot_dir):
        alt = os.path.join(lama_dir, dataset.replace("-", "_"), f"{shot}_shot")
        shot_dir = alt if os.path.isdir(alt) else shot_dir
    files = sorted(glob.glob(os.path.join(shot_dir, "*.jpg")))
    mapping = {}
    mp = os.path.join(shot_dir, "category_mapping.json")
    if os.path.exists(mp):
        try:
            with open(mp) as f:
                mapping = json.load(f)
        except Exception as e:
            print(f"加载类别映射文件时出错: {e}")
    s2i = {os.path.splitext(os.path.basename(p))[0]: p for p in files}
    return s2i, {s: mapping.get(s, s) for s in s2i}


def visualize_results(query_image_path, result_image_paths, output_path, cell=(300, 225)):
    """the per-query contact sheet of the reference (:354-393): query + up to 11 results on a 3 x 4 grid with "Query Image" /
    "Top k" captions.  Drawn with PIL (the reference uses matplotlib; the file is for eyeballing, nothing reads it)."""
    from PIL import Image, ImageDraw
    cw, ch = cell
    sheet = Image.new("RGB", (4 * cw, 3 * (ch + 18)), "white")
    draw = ImageDraw.Draw(sheet)
    items = [("Query Image", query_image_path)] + [(f"Top {i + 1}", p) for i, p in enumerate(result_image_paths[:11])]
    for k, (title, path) in enumerate(items):
        try:
            im = H.load_image_rgb(R.clean_image_path(path))
            im.thumbnail((cw - 8, ch - 8))
        except Exception as e:
            print(f"可视化图像时出错 {path}: {e}")
            continue
        x0, y0 = (k % 4) * cw, (k // 4) * (ch + 18)
        draw.text((x0 + 4, y0 + 2), title, fill="black")
        sheet.paste(im, (x0 + (cw - im.width) // 2, y0 + 18 + (ch - im.height) // 2))
    sheet.save(output_path, quality=85)


def retrieve_dataset(args, dataset, shot, model, preprocess, stem, feats, paths, results_dir, lama_dir, style_cache):
    """ref :773-898"""
    from PIL import Image
    s2i, s2c = get_inpainted_images(lama_dir, dataset, shot)
    if not s2i:
        return None
    cat2s: dict = {}
    for s, c in s2c.items():
        cat2s.setdefault(c, []).append(s)
    # query embeddings in one batch; also the cache files of the reference (:794-822)
    ids = list(s2i)
    tens, ok_ids = [], []
    for s in ids:
        try:
            tens.append(preprocess(Image.open(R.clean_image_path(s2i[s])).convert("RGB")))
            ok_ids.append(s)
        except Exception as e:
            print(f"提取CLIP特征时出错: {e}, 图像: {s2i[s]}")
    qf = R.embed_images(model, torch.stack(tens), args.embed_batch).cpu().numpy() if tens else np.zeros((0, 512), np.float32)
    np.save(os.path.join(results_dir, f"{dataset}_{shot}_shot_inpainted_clip_features.npy"), qf)
    with open(os.path.join(results_dir, f"{dataset}_{shot}_shot_inpainted_image_paths.json"), "w") as f:
        json.dump([s2i[s] for s in ok_ids], f)
    qmap = {s: qf[i] for i, s in enumerate(ok_ids)}
    all_results: dict = {}
    for cat, samples in cat2s.items():
        cat_res = []
        for s in samples:
            if s not in qmap:
                continue
            first = R.clip_first_stage_retrieval(qmap[s], feats, paths, top_k=args.clip_top_k, device=model.device)
            if not first:
                continue
            final = R.resnet_second_stage_rerank(s2i[s], first, stem, style_cache)
            if not final:
                continue
            with open(os.path.join(results_dir, f"{dataset}_{shot}_shot_{cat}_{s}_retrieval_results.json"), "w", encoding="utf-8") as f:
                json.dump(final, f, indent=2, ensure_ascii=False)
            if not args.no_visuals:
                visualize_results(s2i[s], [r["image_path"] for r in final[:10]],
                                  os.path.join(results_dir, f"{dataset}_{shot}_shot_{cat}_{s}_visual.jpg"))
            cat_res.append({"sample_id": s, "image_path": s2i[s], "category": cat, "similar_images": final})
        if cat_res:
            all_results.setdefault(cat, []).extend(cat_res)
    with open(os.path.join(results_dir, f"{dataset}_{shot}_shot_retrieval_results.json"), "w", encoding="utf-8") as f:
        json.dump(all_results, f, indent=2, ensure_ascii=False)
    return all_results


def style_resize_mode(choice: str) -> str:
    """'cv2' or 'restated' for the style re-rank's 256x256 resize.  With OpenCV installed the reference's own calls decide the final
    ranks (near-tied candidates can order differently under the restated, unpinned resize), so 'auto' keeps them; the batched GPU
    route is the default only where OpenCV does not exist, and opt-in (--style-resize restated) where it does."""
    if choice == "restated":
        return "restated"
    try:
        import cv2  # noqa: F401
        return "cv2"
    except ImportError:
        if choice == "cv2":
            raise SystemExit("--style-resize cv2: OpenCV (cv2) is not importable in this environment")
        return "restated"


def main(argv=None):
    args = build_parser().parse_args(argv)
    results_dir = args.output_dir or RESULTS_DIR
    lama_dir = args.lamainpaint_dir or LAMAINPAINT_DIR
    os.makedirs(results_dir, exist_ok=True)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu_id)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        from ..rccl import init_rccl
        init_rccl(device)
    rank = dist.get_rank() if world > 1 else 0
    print(f"使用设备: {device}")
    clip_w, resnet_w = resolve_weights(args)
    model, preprocess = R.load_clip("ViT-B/32", device, weights=clip_w, precision=args.clip_precision)
    if not args.host_preprocess:
        preprocess = R.load_clip_device_preprocess(device)
    stem = R.StemStyle(torch.load(resnet_w, map_location="cpu") if resnet_w else None, device)
    stem.force_restated = style_resize_mode(args.style_resize) == "restated"
    stem.gpu_files = args.decode == "gpu" and stem.force_restated
    feats, paths = {}, {}
    if args.dataset_source in ("coco", "both"):
        f, p = load_or_compute_features(args, "coco", args.coco_dir, None, args.pretrained_coco_features, args.pretrained_coco_paths,
                                        model, preprocess, results_dir, rank == 0)
        if f is not None and len(f):
            feats["coco"], paths["coco"] = f, p
    if args.dataset_source in ("mini-imagenet", "both"):
        f, p = load_or_compute_features(args, "mini_imagenet", args.mini_imagenet_dir, None, args.pretrained_mini_imagenet_features,
                                        args.pretrained_mini_imagenet_paths, model, preprocess, results_dir, rank == 0)
        if f is not None and len(f):
            feats["mini-imagenet"], paths["mini-imagenet"] = f, p
    if not feats:
        print("错误：没有可用的数据集特征，无法进行检索")
        return 1
    all_shots: dict = {}
    style_cache: dict = {}
    sc_path = args.style_cache or os.path.join(results_dir, "style_cache.npz")
    sc_meta = json.dumps(style_cache_meta(stem), sort_keys=True)
    if os.path.exists(sc_path):   # the reference recomputes all 100 candidates' style vectors for every query (:468-470)
        z = np.load(sc_path, allow_pickle=False)
        if "meta" in z.files and str(z["meta"]) == sc_meta:
            style_cache = {p: f for p, f in zip(z["paths"].tolist(), z["feats"])}
            print(f"已加载 {len(style_cache)} 个缓存的风格特征: {sc_path}")
        else:     # other stem weights or another resize backend produced these vectors: mixing them into the L2 re-rank is wrong
            print(f"风格特征缓存与当前ResNet权重/缩放实现不符，已忽略: {sc_path}")
    if rank == 0:   # queries are few; the corpus embedding above is the sharded part
        for ds in args.datasets:
            all_shots[ds] = {}
            for shot in args.shots:
                print(f"\n====== 处理数据集: {ds}, {shot}_shot ======")
                res = retrieve_dataset(args, ds, shot, model, preprocess, stem, feats, paths, results_dir, lama_dir, style_cache)
                if res:
                    all_shots[ds][f"{shot}_shot"] = res
                else:
                    print(f"跳过数据集 {ds} 的 {shot}_shot")
        if style_cache:
            np.savez(sc_path, paths=np.array(list(style_cache), dtype=str), feats=np.stack(list(style_cache.values())),
                     meta=np.array(sc_meta))
        if any(all_shots.values()):
            out = os.path.join(results_dir, "all_shots_retrieval_results.json")
            with open(out, "w", encoding="utf-8") as f:
                json.dump(all_shots, f, indent=2, ensure_ascii=False)
            print(f"所有数据集和所有shot的检索结果已合并保存到 {out}")
        else:
            print("没有成功检索任何数据集")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
