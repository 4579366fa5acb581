"""Stage 0 — drop-in for ``lama_inpaint/lama_inpaint.py`` (flags :227-233, paths :82-99, loop :139-215): erase the annotated
objects of every k-shot training image with LaMa and write ``../lamainpaint/<dataset>/<k>_shot/<file_name>``.

    cd lama_inpaint && python -m domain_rag_amd.cli.stage0_lama --datasets ArTaxOr --shots 1

The generator runs on the HIP path (domain-rag_amd/lama.py); the model is loaded once for all datasets.  Under
torch.distributed.run (RANK / WORLD_SIZE) the images of a dataset are sharded with the reference's contiguous rule — no
collective is needed, every rank writes its own files and logs its own counters.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import time
from collections import defaultdict
from datetime import datetime

from .. import hostlogic as H


def setup_logger():
    log_dir = "../lamainpaint/logs"
    os.makedirs(log_dir, exist_ok=True)
    stamp = os.environ.get("DRAG_TIMESTAMP") or datetime.now().strftime("%Y%m%d_%H%M%S")
    rank = os.environ.get("RANK")
    log_file = os.path.join(log_dir, f"lama_inpaint_{stamp}{'_rank' + rank if rank else ''}.log")
    logger = logging.getLogger("domain_rag_amd.lama")
    logger.setLevel(logging.INFO)
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    for h in (logging.FileHandler(log_file), logging.StreamHandler()):
        h.setFormatter(fmt)
        logger.addHandler(h)
    logger.propagate = False
    return logger


def build_parser():
    p = argparse.ArgumentParser(description="LaMa Inpainting for Multiple Datasets (MI355X)")
    p.add_argument("--datasets", nargs="+", default=["ArTaxOr", "clipart1k", "DIOR", "FISH", "NEU-DET"], help="要处理的数据集列表")
    p.add_argument("--shots", nargs="+", default=["1", "2", "3", "5", "10"], help="每个数据集要处理的shot数量")
    p.add_argument("--fix-channels", action="store_true", help="修复通道不匹配问题")          # parsed and unused in the reference too
    # additions
    p.add_argument("--lama-model", type=str, default=None, help="big-lama.pt (default: $LAMA_MODEL or ./model/big-lama.pt)")
    p.add_argument("--synthetic-weights", action="store_true")
    p.add_argument("--tiny", action="store_true", help="test hook: tiny generator")
    return p


def _annotation_index(annotation_file):
    """-> (image_id -> {file_name, width, height}, image_id -> [annotations] in file order, category_id -> name)"""
    with open(annotation_file, "r") as f:
        data = json.load(f)
    info_of = {im["id"]: {k: im[k] for k in ("file_name", "width", "height")} for im in data.get("images", [])}
    anns_of = defaultdict(list)
    for ann in data.get("annotations", []):
        anns_of[ann["image_id"]].append(ann)
    names = {c["id"]: c["name"] for c in data.get("categories", [])}
    return info_of, anns_of, names, len(data.get("annotations", []))


def _inpaint_one(simple_lama, image_path, info, boxes):
    """one image of the loop (:159-215): RGB, annotated size (PIL's default bicubic resize if the file differs), union mask"""
    from PIL import Image
    image = Image.open(image_path)
    if image.mode != "RGB":
        image = image.convert("RGB")
    size = (info["width"], info["height"])
    if image.size != size:
        image = image.resize(size)
    mask = Image.fromarray(H.inpaint_mask_array(size[0], size[1], boxes), mode="L")
    return simple_lama(image, mask)


def process_dataset(dataset_name, shot_count, logger, simple_lama, rank: int = 0, world: int = 1):
    """process_dataset (:78-224) -> (processed, errors).  ``simple_lama`` is the model object: (PIL RGB, PIL L) -> PIL."""
    logger.info(f"数据集 {dataset_name} / {shot_count}-shot: 开始")
    dataset_path = os.path.join("../datasets", dataset_name)
    train_images_dir = os.path.join(dataset_path, "train")
    for need, what in ((dataset_path, "数据集目录"), (train_images_dir, "训练图像目录")):
        if not os.path.exists(need):
            logger.error(f"{what}不存在: {need}")
            return 0, 0
    annotation_file = os.path.join(dataset_path, "annotations", f"{shot_count}_shot.json")
    output_dir = H.lama_output_dir(dataset_name, shot_count)
    os.makedirs(output_dir, exist_ok=True)
    logger.info(f"结果写入 {output_dir}")
    try:
        info_of, anns_of, names, n_ann = _annotation_index(annotation_file)
    except Exception as e:
        logger.error(f"读取注释文件 {annotation_file} 失败: {e}")
        return 0, 0
    logger.info(f"注释文件 {annotation_file}: {len(info_of)} 个图像, {n_ann} 个注释, {len(anns_of)} 个图像带有bbox")
    work = list(anns_of.items())                                  # annotation order (dict insertion), like the reference's loop
    if world > 1:
        work = H.split_samples_for_gpus(work, world)[rank]
    done = failed = multi = 0
    for image_id, anns in work:
        info = info_of.get(image_id)
        if info is None:
            logger.warning(f"注释引用了未知的图像ID {image_id}, 跳过")
            continue
        image_path = os.path.join(train_images_dir, info["file_name"])
        if len(anns) > 1:
            multi += 1
            cats = ", ".join("{}(ID:{})".format(names.get(a["category_id"], "未知类别 {}".format(a["category_id"])), a["category_id"]) for a in anns)
            logger.info(f"多bbox图像 {info['file_name']}: {len(anns)} 个bbox, 类别: {cats}")
        try:
            result = _inpaint_one(simple_lama, image_path, info, [a["bbox"] for a in anns])
            out_name = os.path.join(output_dir, info["file_name"])
            os.makedirs(os.path.dirname(out_name), exist_ok=True)
            result.save(out_name)
            done += 1
        except Exception as e:                                     # reference convention: log, count, continue
            logger.error(f"图像 {image_path} 处理失败: {e}")
            failed += 1
    logger.info(f"数据集 {dataset_name} / {shot_count}-shot: 完成 {done} 个图像, 失败 {failed} 个, 其中多bbox图像 {multi} 个")
    return done, failed


def main(argv=None):
    args = build_parser().parse_args(argv)
    logger = setup_logger()
    logger.info(f"LaMa stage: 数据集 {', '.join(args.datasets)}; shots {', '.join(args.shots)}")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    if args.synthetic_weights:
        os.environ["DRAG_SYNTHETIC_WEIGHTS"] = "1"
    if args.tiny:
        os.environ["DRAG_TINY"] = "1"
    if args.lama_model:
        os.environ["LAMA_MODEL"] = args.lama_model
    from ..lama import SimpleLama
    simple_lama = SimpleLama()                                     # once, not per dataset (:104) — same outputs
    t_start = time.time()
    totals = [0, 0]
    for ds in args.datasets:
        for shot in args.shots:
            try:
                for i, v in enumerate(process_dataset(ds, shot, logger, simple_lama, rank, world)):
                    totals[i] += v
            except Exception as e:
                logger.error(f"数据集 {ds} / {shot}-shot 中断: {e}")
    elapsed = time.time() - t_start
    logger.info(f"全部完成: {totals[0]} 个图像成功, {totals[1]} 个失败, 用时 {int(elapsed // 3600)}:{int(elapsed % 3600 // 60):02d}:{elapsed % 60:05.2f}")
    return tuple(totals)


if __name__ == "__main__":
    main()
