"""Stage 2 — drop-in for ``batch_generate_flux_kshot.py`` (flags :33-45; outputs :798-802,:477-519).

    python -m domain_rag_amd.cli.stage2_generate --dataset ArTaxOr --shots 1 --retrieval_results_dir ./retrieval/retrieval_results

For every k-shot sample: ranks 1..5 of ``all_shots_retrieval_results.json`` -> Redux prior over [retrieved, target]
with scales [0.8, 1.0] -> FLUX.1-dev 1024x1024, 50 steps, guidance 2.5, seed 0 -> ``generated_image_rank{r}.png`` plus
the side files stage 3 and humans read.  Under torch.distributed.run the (sample, rank) units are sharded with the
reference's contiguous rule; no collective is needed.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import shutil
from datetime import datetime

import torch

from ..engine import Engine, generator_noise, pack_noise
from .. import hostlogic as H
from ..retrieval import shard_bounds

COCO_IMAGE_SCALE, TARGET_IMAGE_SCALE, COCO_TEXT_SCALE, TARGET_TEXT_SCALE = 0.8, 1.0, 1.0, 1.0
PROMPT = ""
GUIDANCE, STEPS, SIZE, SEED = 2.5, 50, 1024, 0


def build_parser():
    p = argparse.ArgumentParser(description="批量生成Flux k-shot图像 (MI355X)")
    p.add_argument("--dataset", type=str, default=None)
    p.add_argument("--shots", nargs="+", type=int, default=None)
    p.add_argument("--output_dir", type=str, default="result")
    p.add_argument("--database", type=str, default="coco", choices=["coco", "miniimagenet", "neudet"])
    p.add_argument("--retrieval_results_dir", type=str, default="./retrieval/retrieval_results")
    p.add_argument("--dataset_group", type=str, default=None, choices=["dataset1", "dataset2", "dataset3", "dataset4"])
    # additions
    p.add_argument("--lamainpaint_dir", type=str, default="./lamainpaint")
    p.add_argument("--model_root", type=str, default="./model")
    p.add_argument("--coco_dir", type=str, default="./retrieval/coco/train2017",
                   help="the reference's hard-coded coco_dataset_dir (batch_…:29)")
    p.add_argument("--synthetic-weights", action="store_true")
    p.add_argument("--tiny", action="store_true", help="test hook: tiny architectures, small images")
    p.add_argument("--num_inference_steps", type=int, default=STEPS)
    p.add_argument("--size", type=int, default=SIZE)
    p.add_argument("--io_workers", type=int, default=4, help="background PNG encoder processes (0 = write inline like the reference)")
    p.add_argument("--png", choices=["gpu", "host"], default="gpu",
                   help="generated_image_rank{r}.png: 'gpu' = encoded on the device (domain_rag_amd.png: same pixels, not Pillow's bytes), "
                        "'host' = Pillow (zlib level 6) in the --io_workers processes")
    p.add_argument("--ref_batch", type=int, default=8, help="references of one target generated per batch (1 = one at a time like the reference)")
    p.add_argument("--fallback-seed", type=int, default=None, help="seed the random-COCO fallback (reference: unseeded)")
    return p


DATASET_GROUPS = {"dataset1": ["ArTaxOr", "clipart1k"], "dataset2": ["DIOR", "FISH"], "dataset3": ["NEU-DET", "UODD"],
                  "dataset4": ["NWPU_VHR_10", "Camouflage"]}


def generate_ranked(engine: Engine, target_path, items, sdir, args, database):
    """generate_image (:439-524) for the retrieved references of ONE target, composited as one batch (the reference
    runs them one at a time; each image's arithmetic is independent of its batch neighbours, so the pixels are the same).
    ``items`` = [(similarity, ref_path, rank)].  Returns (n_ok, n_failed); failures are logged and skipped."""
    from PIL import Image
    ok = bad = 0
    try:
        tgt_img = H.load_image_rgb(target_path)
    except Exception as e:   # reference convention: log, continue
        print(f"生成图像时出错: {str(e)}")
        return 0, len(items)
    w, h = tgt_img.size
    w, h = max((w // 16) * 16, 64), max((h // 16) * 16, 64)        # computed and recorded, not used (:447-456,:470-471)
    ready = []
    for sim, ref_path, rank in items:
        try:
            ref_img = H.load_image_rgb(ref_path)
            ready.append((sim, ref_path, rank, engine.prior_embeds([ref_img, tgt_img], PROMPT, [COCO_IMAGE_SCALE, TARGET_IMAGE_SCALE],
                                                                   [COCO_TEXT_SCALE, TARGET_TEXT_SCALE])))
        except Exception as e:
            print(f"生成图像时出错: {str(e)}")
            bad += 1
    bs = max(1, args.ref_batch)
    for c0 in range(0, len(ready), bs):
        chunk = ready[c0:c0 + bs]
        try:
            pe, pp = torch.cat([c[3][0] for c in chunk], 0), torch.cat([c[3][1] for c in chunk], 0)
            noise = pack_noise(generator_noise(SEED, 1, args.size, args.size, 1)[0]).expand(len(chunk), -1, -1).contiguous()   # seed 0 every time (:468)
            imgs_dev = engine.pipe(pe, pp, height=args.size, width=args.size, guidance_scale=GUIDANCE,
                                   num_inference_steps=args.num_inference_steps, noise_tokens=noise)
            if getattr(args, "png", "host") == "gpu":       # whole .png files come back instead of raw pixels
                from .. import png as gpu_png
                imgs = gpu_png.encode(imgs_dev)
            else:
                imgs = imgs_dev.cpu().numpy()
        except Exception as e:
            print(f"生成图像时出错: {str(e)}")
            bad += len(chunk)
            continue
        os.makedirs(sdir, exist_ok=True)
        for (similarity, ref_path, rank, _), arr in zip(chunk, imgs):
            try:
                img_path = os.path.join(sdir, f"generated_image_rank{rank}.png")
                writer = getattr(args, "writer", None)
                if isinstance(arr, bytes):
                    with open(img_path, "wb") as f:
                        f.write(arr)
                elif writer is None:
                    Image.fromarray(arr).save(img_path)
                else:
                    writer.save(Image.fromarray(arr), img_path)          # PNG encode off the generation thread
                pf = os.path.join(sdir, "params.txt")
                if not os.path.exists(pf):
                    with open(pf, "w") as f:
                        f.write(f"数据库类型: {database}\n参考图像权重: {COCO_IMAGE_SCALE}\n目标图像权重: {TARGET_IMAGE_SCALE}\n"
                                f"参考文本权重: {COCO_TEXT_SCALE}\n目标文本权重: {TARGET_TEXT_SCALE}\n提示词: {PROMPT}\n指导比例: {GUIDANCE}\n"
                                f"推理步数: {args.num_inference_steps}\n生成图像尺寸: {w}x{h}\n原始图像尺寸: {tgt_img.size[0]}x{tgt_img.size[1]}\n")
                rs = f"rank{rank}" if rank is not None else ""
                ss = f"_sim{similarity:.4f}" if similarity is not None else ""
                with open(os.path.join(sdir, f"ref_info{rs}{ss}.txt"), "w") as f:
                    f.write(f"数据库类型: {database}\n参考图像: {ref_path}\n目标图像: {target_path}\n生成图像尺寸: {w}x{h}\n"
                            f"原始图像尺寸: {tgt_img.size[0]}x{tgt_img.size[1]}\n")
                    if rank is not None:
                        f.write(f"排名: {rank}\n")
                    if similarity is not None:
                        f.write(f"相似度: {similarity}\n")
                tgt_out = os.path.join(sdir, "target_input.png")
                if not os.path.exists(tgt_out):
                    shutil.copy(target_path, tgt_out)
                shutil.copy(ref_path, os.path.join(sdir, f"ref_input{rs}.jpg"))
                ok += 1
            except Exception as e:
                print(f"生成图像时出错: {str(e)}")
                bad += 1
    return ok, bad


def process_dataset(engine, results, dataset, shot, args, rank, world):
    """process_kshot_dataset_with_retrieval (:766-1058)"""
    shot_dir = os.path.join(args.lamainpaint_dir, dataset, f"{shot}_shot")
    if not os.path.isdir(shot_dir):
        alt = os.path.join(args.lamainpaint_dir, dataset.replace("-", "_"), f"{shot}_shot")
        if not os.path.isdir(alt):
            print(f"错误：找不到k-shot目录 {shot_dir}")
            return 0, 0
        shot_dir = alt
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(shot_dir) if f.endswith(".jpg"))
    if not names:
        print(f"跳过数据集 {dataset} {shot}-shot，因为找不到样本")
        return 0, 0
    result_dir = f"{args.output_dir}/{dataset}_{shot}shot_retrieval"
    ts = getattr(args, "run_ts", None) or run_timestamp(world)       # several ranks: the run's one stamp (main); one process: now
    base = os.path.join(result_dir, f"results_coco_{COCO_IMAGE_SCALE}_target_{TARGET_IMAGE_SCALE}_cocotext_{COCO_TEXT_SCALE}"
                                    f"_targettext_{TARGET_TEXT_SCALE}_{ts}")
    os.makedirs(base, exist_ok=True)
    if rank == 0:
        with open(os.path.join(base, "batch_params.txt"), "w") as f:
            f.write(f"数据集: {dataset} ({shot}-shot，使用检索结果)\nCOCO图像权重: {COCO_IMAGE_SCALE}\n目标图像权重: {TARGET_IMAGE_SCALE}\n"
                    f"COCO文本权重: {COCO_TEXT_SCALE}\n目标文本权重: {TARGET_TEXT_SCALE}\n提示词: {PROMPT}\n指导比例: {GUIDANCE}\n"
                    f"推理步数: {args.num_inference_steps}\n处理样本数: {len(names)}\n"
                    f"为每个样本生成: 最多10张图像 (基于相似度最高的COCO图像)\n图像尺寸: 动态调整至与目标图像匹配 (保证是16的倍数)\n")
    rng = random.Random(args.fallback_seed)
    s, e = shard_bounds(len(names), world, rank)
    ok = bad = images = 0            # samples that produced at least one image / samples that produced none / images written
    stamp = lambda: datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    for name in names[s:e]:
        target = os.path.join(shot_dir, name + ".jpg")
        sdir = os.path.join(base, name)
        os.makedirs(sdir, exist_ok=True)
        try:
            top = H.top5_similar_images(results, name, dataset, shot, args.coco_dir, rng)
            if not top:                                  # (:946-956)
                print(f"跳过样本 {name}，因为找不到相似图像")
                with open(os.path.join(sdir, "error.txt"), "w") as f:
                    f.write(f"处理样本 {name} 时出错: 找不到相似图像\n时间: {stamp()}\n")
                bad += 1
                continue
        except ValueError as ex:                         # NEU-DET / COCO finders have no random fallback (:957-973)
            print(f"处理样本 {name} 时出错: {str(ex)}")
            with open(os.path.join(sdir, "error.txt"), "w") as f:
                f.write(f"处理样本 {name} 时出错: {str(ex)}\n时间: {stamp()}\n")
            bad += 1
            continue
        top = [t for t in top if t[2] <= 5]
        n_ok, _ = generate_ranked(engine, target, top, sdir, args, args.database)   # paths are already fixed up and checked
        images += n_ok
        if n_ok:
            ok += 1
        else:                                            # (:1033-1044)
            bad += 1
            with open(os.path.join(sdir, "generation_failed.txt"), "w") as f:
                f.write(f"生成样本 {name} 的图像失败\n时间: {stamp()}\n")
                if top:
                    f.write(f"找到了 {len(top)} 个相似图像，但生成全部失败\n")
                    for sim, path, r in top:
                        f.write(f"  - Rank {r}: {path} (相似度: {sim:.4f})\n")
    for path, err in (args.writer.flush() if getattr(args, "writer", None) is not None else []):
        print(f"生成图像时出错: 保存 {path} 失败: {err}")
        images -= 1
    # the closing summary (:1046-1056); the size statistics read ref_info{rank}.txt, which is never written under that name
    # (the file is ref_inforank{r}_sim….txt), so the list stays empty in the reference too
    with open(os.path.join(base, "batch_params.txt" if world == 1 else f"batch_params_rank{rank}.txt"), "a") as f:
        f.write(f"成功处理样本数: {ok}\n失败处理样本数: {bad}\n总共生成图像数: {images}\n\n生成图像尺寸统计:\n\n完成时间: {stamp()}\n")
    print(f"数据集 {dataset} {shot}-shot处理完成：成功 {ok} 个样本，失败 {bad} 个样本，总共生成 {images} 张图像")
    return ok, bad


def run_timestamp(world: int) -> str:
    """the ``_<timestamp>`` suffix of a results directory (batch_generate_flux_kshot.py:798-802).  One process: now, like the
    reference.  Several ranks of one launch must agree on ONE directory per dataset (each rank calling now() splits a dataset run
    over directories whenever the ranks cross a second boundary): $DRAG_TIMESTAMP if the launcher exports one; else rank 0's
    clock, BROADCAST over a gloo process group (works across nodes and under any launcher that provides RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT — torchrun, srun + env, mpirun wrappers; the data path itself needs no collective, the group exists
    for this one string); only if no rendezvous is POSSIBLE (no MASTER_ADDR / MASTER_PORT, or the group cannot be created), the start
    time of the common parent process read from /proc (same node, same parent only — the warning says so).  Once a group exists a
    failing broadcast is an error, never a silent fall-back: ranks that disagree about the directory corrupt the run.
    main() calls this ONCE, before any work, while the ranks are still in step, and every (dataset, shot) of the run re-uses the
    stamp (directory names already carry dataset and shot): a collective per dataset would be reached minutes apart by ranks whose
    shards differ in cost."""
    env = os.environ.get("DRAG_TIMESTAMP")
    if env:
        return env
    if world > 1:
        import torch.distributed as dist
        created = False
        if not dist.is_initialized():
            try:
                if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
                    raise RuntimeError("MASTER_ADDR / MASTER_PORT are not set")
                # a rank that never arrives must not hold the others for gloo's 30-minute default ($DRAG_RENDEZVOUS_TIMEOUT_S to change)
                from datetime import timedelta
                dist.init_process_group("gloo", timeout=timedelta(seconds=int(os.environ.get("DRAG_RENDEZVOUS_TIMEOUT_S", "300"))))
                created = True
            except Exception as e:
                print(f"警告：无法建立进程组来广播时间戳 ({e})；退回到父进程启动时间（仅限同一节点、同一父进程）")
        if dist.is_initialized():
            box = [datetime.now().strftime("%Y%m%d_%H%M%S")]
            try:
                dist.broadcast_object_list(box, src=0)
            finally:
                if created:           # the group existed for this one string: the data path needs no collective
                    dist.destroy_process_group()
            return box[0]
        try:
            with open(f"/proc/{os.getppid()}/stat") as f:
                ticks = int(f.read().rsplit(")", 1)[1].split()[19])          # field 22: start time in clock ticks since boot
            with open("/proc/stat") as f:
                btime = next(int(ln.split()[1]) for ln in f if ln.startswith("btime"))
            return datetime.fromtimestamp(btime + ticks / os.sysconf("SC_CLK_TCK")).strftime("%Y%m%d_%H%M%S")
        except Exception as e:
            print(f"警告：无法确定共享时间戳 ({e})；各进程的结果目录可能不同，请设置 DRAG_TIMESTAMP")
    return datetime.now().strftime("%Y%m%d_%H%M%S")


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    # several ranks: ONE results-directory stamp for the whole run, agreed here — before the weights load and before any rank can
    # run ahead of the others; one process keeps the reference's now() per dataset
    args.run_ts = run_timestamp(world) if world > 1 else None
    if args.tiny and args.size == SIZE:
        args.size = 64
    rf = os.path.join(args.retrieval_results_dir, "all_shots_retrieval_results.json")
    if not os.path.exists(rf):
        print(f"错误：找不到检索结果文件 {rf}")
        return 1
    with open(rf, encoding="utf-8") as f:
        results = json.load(f)
    from ..io_pool import ImageWriter
    args.writer = ImageWriter(args.io_workers)
    engine = Engine("dev", args.model_root, synthetic=args.synthetic_weights, tiny=args.tiny, device=torch.device("cuda", local))
    datasets = [args.dataset] if args.dataset else (DATASET_GROUPS[args.dataset_group] if args.dataset_group else list(results))
    tot_ok = tot_bad = 0
    for ds in datasets:
        shots = args.shots or sorted(int(k.split("_")[0]) for k in results.get(ds, {}) if k.endswith("_shot"))
        for shot in shots:
            ok, bad = process_dataset(engine, results, ds, shot, args, rank, world)
            tot_ok += ok
            tot_bad += bad
    args.writer.close()
    print(f"处理完成: 成功 {tot_ok} 个样本, 失败 {tot_bad} 个样本")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
