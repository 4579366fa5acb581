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


def process_dataset(dataset_name, shot_count, logger, simple_lama, rank: int = 0, world: int = 1):
    """process_dataset (:78-224) -> (processed, errors).  ``simple_lama`` is the model object: (PIL RGB, PIL L) -> PIL."""
    from PIL import Image
    logger.info(f"开始处理数据集: {dataset_name}, {shot_count}-shot")
    dataset_path = os.path.join("../datasets", dataset_name)
    if not os.path.exists(dataset_path):
        logger.error(f"数据集路径不存在: {dataset_path}")
        return 0, 0
    train_images_dir = os.path.join(dataset_path, "train")
    if not os.path.exists(train_images_dir):
        logger.error(f"训练图像目录不存在: {train_images_dir}")
        return 0, 0
    annotation_file = os.path.join(dataset_path, "annotations", f"{shot_count}_shot.json")
    output_dir = H.lama_output_dir(dataset_name, shot_count)
    os.makedirs(output_dir, exist_ok=True)
    logger.info(f"输出目录: {output_dir}")
    try:
        with open(annotation_file, "r") as f:
            data = json.load(f)
        logger.info(f"成功加载注释文件 {annotation_file}")
    except Exception as e:
        logger.error(f"无法加载注释文件 {annotation_file}: {e}")
        return 0, 0
    info_of = {im["id"]: {"file_name": im["file_name"], "width": im["width"], "height": im["height"]} for im in data.get("images", [])}
    anns_of = defaultdict(list)
    for ann in data.get("annotations", []):
        anns_of[ann["image_id"]].append(ann)
    names = {c["id"]: c["name"] for c in data.get("categories", [])}
    logger.info(f"找到 {len(info_of)} 个图像和 {len(data.get('annotations', []))} 个注释")
    logger.info(f"共有 {len(anns_of)} 个不同的图像需要处理")
    work = list(anns_of.items())                                  # annotation order (dict insertion), like the reference's loop
    if world > 1:
        work = H.split_samples_for_gpus(work, world)[rank]
    processed = errors = multi = 0
    for image_id, anns in work:
        if image_id not in info_of:
            logger.warning(f"警告: 找不到图像ID {image_id} 的信息")
            continue
        info = info_of[image_id]
        image_path = os.path.join(train_images_dir, info["file_name"])
        if len(anns) > 1:
            multi += 1
            cats = ", ".join("{}(ID:{})".format(names.get(a["category_id"], "未知类别 {}".format(a["category_id"])), a["category_id"]) for a in anns)
            logger.info(f"处理多bbox图像: {info['file_name']}, bbox数量: {len(anns)}, 类别: {cats}")
        try:
            image = Image.open(image_path)
            if image.mode != "RGB":
                image = image.convert("RGB")
            if image.width != info["width"] or image.height != info["height"]:
                image = image.resize((info["width"], info["height"]))        # PIL default filter (bicubic), :167
            mask = Image.fromarray(H.inpaint_mask_array(info["width"], info["height"], [a["bbox"] for a in anns]), mode="L")
            result = simple_lama(image, mask)
            out_name = os.path.join(output_dir, info["file_name"])
            os.makedirs(os.path.dirname(out_name), exist_ok=True)
            result.save(out_name)
            processed += 1
        except Exception as e:                                     # reference convention: log, count, continue
            logger.error(f"处理图像 {image_path} 时出错: {e}")
            errors += 1
    logger.info(f"完成数据集 {dataset_name} {shot_count}-shot 的处理: 成功处理 {processed} 个图像, 错误 {errors} 个")
    logger.info(f"其中处理了 {multi} 个有多个bbox的图像")
    return processed, errors


def main(argv=None):
    args = build_parser().parse_args(argv)
    logger = setup_logger()
    logger.info("LaMa Inpainting开始执行")
    logger.info(f"将处理以下数据集: {', '.join(args.datasets)}")
    logger.info(f"将处理以下shot数量: {', '.join(args.shots)}")
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
    start = time.time()
    total_ok = total_bad = 0
    for ds in args.datasets:
        for shot in args.shots:
            try:
                ok, bad = process_dataset(ds, shot, logger, simple_lama, rank, world)
                total_ok += ok
                total_bad += bad
            except Exception as e:
                logger.error(f"处理数据集 {ds} {shot}-shot 时发生错误: {e}")
    total = time.time() - start
    hours, rem = divmod(total, 3600)
    minutes, seconds = divmod(rem, 60)
    logger.info(f"所有数据集处理完成: 成功处理 {total_ok} 个图像, 错误 {total_bad} 个")
    logger.info(f"总运行时间: {int(hours)}小时 {int(minutes)}分钟 {seconds:.2f}秒")
    return total_ok, total_bad


if __name__ == "__main__":
    main()
