"""Stage 3 — drop-in for ``outpainting_updown_sampling_redux.py`` (flags :1894-1911, outputs :1260-1322,:1579-1603,:1813-1886).

    python -m domain_rag_amd.cli.stage3_outpaint --process_id 1 --dataset clipart1k --shot 1 [--multi_gpu --num_gpus 8]

Per sample: original image + all bboxes -> resolution policy -> keep-bbox mask -> for every generated background:
Redux prior(bg) -> Flux-Fill (50 steps x strength) -> ``*_hires_result_*.png`` -> resized back -> ``*_final_result_*.png``
+ ``*_params_*.json``; then ``outpaint_results_<k>shot.json`` and the ``final_results`` collection.  Models are loaded once
per process.  Multi-GPU = one process per GPU over contiguous sample chunks (the reference's ``split_samples_for_gpus``),
launched either by torch.distributed.run (RANK/WORLD_SIZE) or by ``--multi_gpu`` (this CLI re-spawns itself per GPU);
no tensor ever crosses a GPU boundary, results are merged from per-GPU JSON files.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import random
import shutil
import subprocess
import sys
import time
import traceback

import numpy as np
import torch

from .. import hostlogic as H
from ..engine import Engine, generator_noise, pack_noise
from ..io_pool import ImageWriter

RESULT_DIR, DATASETS_DIR = "./result", "./datasets"


def build_parser():
    p = argparse.ArgumentParser(description="高分辨率Outpainting处理脚本 (MI355X)")
    p.add_argument("--dataset", type=str)
    p.add_argument("--dataset_group", type=str, choices=["1", "2", "all"], default="all")
    p.add_argument("--sample_id", type=str)
    p.add_argument("--shot", type=int, default=1, choices=[1, 2, 3, 5, 10, 20])
    p.add_argument("--min_dimension", type=int, default=H.MIN_DIMENSION)       # overridden by the per-dataset table, as in the reference
    p.add_argument("--custom_upscale", type=str)
    p.add_argument("--process_id", type=str)
    p.add_argument("--collect_only", action="store_true")
    p.add_argument("--multi_bbox", action="store_true")                        # inert, as in the reference
    p.add_argument("--resume", action="store_true")
    p.add_argument("--log_file", type=str)
    p.add_argument("--failed_only", action="store_true")
    p.add_argument("--multi_gpu", action="store_true")
    p.add_argument("--num_gpus", type=int)
    # additions
    p.add_argument("--seed", type=int, default=None, help="base seed (reference: unseeded random.randint per image)")
    p.add_argument("--model_root", type=str, default="./model")
    p.add_argument("--result_dir", type=str, default=RESULT_DIR)
    p.add_argument("--datasets_dir", type=str, default=DATASETS_DIR)
    p.add_argument("--synthetic-weights", action="store_true")
    p.add_argument("--tiny", action="store_true", help="test hook: tiny architectures, min_dimension 64")
    p.add_argument("--num_inference_steps", type=int, default=H.NUM_INFERENCE_STEPS)
    p.add_argument("--io_workers", type=int, default=4, help="background PNG encoder processes (0 = write inline like the reference)")
    p.add_argument("--png", choices=["gpu", "host"], default="gpu",
                   help="*_hires_result_* / *_final_result_*: 'gpu' = encoded on the device (domain_rag_amd.png: same pixels, not "
                        "Pillow's bytes), 'host' = Pillow (zlib level 6) in the --io_workers processes; masks and copies stay with Pillow")
    p.add_argument("--bg_batch", type=int, default=8, help="backgrounds of one sample composited per batch (1 = one at a time like the reference)")
    return p


def dataset_result_dirs(result_dir, dataset, shot):
    """get_dataset_results (:795-826) incl. the NWPU_VHR-10 -> NWPU_VHR_10 directory rename"""
    name = "NWPU_VHR_10" if dataset == "NWPU_VHR-10" else dataset
    base = os.path.join(result_dir, f"{name}_{shot}shot_retrieval")
    if not os.path.isdir(base):
        print(f"警告：找不到数据集 {dataset} 的 {shot}shot 结果目录: {base}")
        return []
    return sorted(os.path.join(base, f) for f in os.listdir(base) if f.startswith("results_") and os.path.isdir(os.path.join(base, f)))


def find_sample_dirs(result_dir, dataset, shot):
    """sample id -> directory that holds target_input.png + generated_image*.png (:219-239)"""
    out = {}
    for rd in dataset_result_dirs(result_dir, dataset, shot):
        for s in sorted(os.listdir(rd)):
            d = os.path.join(rd, s)
            if os.path.isdir(d) and glob.glob(os.path.join(d, "generated_image*png")):
                out[s] = d
    return out


def parse_resume_log(path):
    """--resume / --failed_only (:1953-1993): the two Chinese log lines are the contract"""
    done, failed = set(), set()
    if path and os.path.exists(path):
        for line in open(path, encoding="utf-8", errors="ignore"):
            if "处理完成" in line and "样本" in line:
                done.add(line.split("样本")[1].split("处理完成")[0].strip())
            elif "处理失败" in line and "样本" in line:
                failed.add(line.split("样本")[1].split("处理失败")[0].strip())
    return done, failed


def process_sample(engine: Engine, args, dataset, sample_id, sample_dir, shot, process_id, rng, writer=None):
    """process_sample_hires (:872-1361).  ``writer``: io_pool.ImageWriter for the large PNGs (None = write inline)"""
    from PIL import Image
    save = (lambda im, path: im.save(path)) if writer is None else writer.save
    t0 = time.time()
    prefix = f"{dataset}_{sample_id}_{shot}shot"
    log = {"sample_id": sample_id, "sample_prefix": prefix, "status": "processing", "outpainted_images": [], "shot_number": shot}
    try:
        ann = H.load_annotation_file(args.datasets_dir, dataset, shot)
        found = H.lookup_sample_annotations(ann, sample_id) if ann else None
        if not found:
            raise RuntimeError(f"找不到与样本ID {sample_id} 匹配的图像/注释")
        info, bboxes, cats = found
        img_path = os.path.join(args.datasets_dir, dataset, "train", info["file_name"])
        original = H.load_image_rgb(img_path)
        crops = [H.crop_box(b, *original.size) for b in bboxes]
        log.update(image_id=info["id"] if info.get("id") else "unknown", categories=cats, category=cats[0] if cats else "unknown",
                   original_resolution=list(original.size), original_image_size=list(original.size), bbox_coords_list=bboxes,
                   bbox_image_sizes=[[c[2] - c[0], c[3] - c[1]] for c in crops])
        out_dir = os.path.join(f"./outpaint_hires/process_{process_id}", dataset, f"{shot}_shot", sample_id)
        os.makedirs(out_dir, exist_ok=True)
        orig_saved = os.path.join(out_dir, f"{prefix}_original.png")
        save(original, orig_saved)
        bbox_saved = []                                   # every bbox crop next to the results (:1116-1131)
        for i, c in enumerate(crops):
            bp = os.path.join(out_dir, f"{prefix}_bbox{i + 1}_original.jpg")
            try:
                original.crop(c).save(bp)
                bbox_saved.append(bp)
            except Exception as e:
                print(f"保存bbox图像{i + 1}失败: {str(e)}")
                bbox_saved.append(None)
        log["bbox_saved_paths"] = bbox_saved
        bgs = sorted(glob.glob(os.path.join(sample_dir, "generated_image*png")))
        if not bgs:
            raise RuntimeError(f"在 {sample_dir} 中找不到生成的背景图像")
        min_dim = 64 if args.tiny else H.UPSCALE_DIMENSION.get(dataset, args.min_dimension)
        try:
            processed, up, down, wu, wd = H.process_image_resolution(original, min_dim, H.MAX_DIMENSION)
        except ValueError as e:
            raise ValueError(f"样本 {sample_id} 处理失败: {str(e)}")
        log.update(upscaled_resolution=list(processed.size), up_scale_factor=up, down_scale_factor=down, was_upscaled=wu,
                   was_downscaled=wd, min_dimension_used=min_dim)
        if wd:
            save(processed, os.path.join(out_dir, f"{prefix}_downscaled_bg.png"))
            log["downscaled_resolution"] = list(processed.size)
        if wu:
            save(processed, os.path.join(out_dir, f"{prefix}_upscaled_bg.png"))
        pb = H.scale_bboxes(bboxes, up, down, wu, wd)
        mask_img, _ = H.generate_outpaint_mask(processed, pb)
        strength = H.STRENGTH.get(dataset, H.DEFAULT_STRENGTH)
        guidance = H.GUIDANCE_SCALE.get(dataset, H.DEFAULT_GUIDANCE)
        ips = H.IMAGE_PROMPT_SCALE.get(dataset, 1.0)
        prompt = H.REDUX_PROMPT.get(dataset, "")
        # FluxFillPipeline works on multiples of 16 (image_processor resizes to them)
        Wp, Hp = processed.size
        W16, H16 = max(Wp // 16 * 16, 16), max(Hp // 16 * 16, 16)
        im16 = processed if (W16, H16) == (Wp, Hp) else processed.resize((W16, H16), Image.LANCZOS)
        mk16 = mask_img if (W16, H16) == (Wp, Hp) else mask_img.resize((W16, H16), Image.LANCZOS)
        img_u8 = torch.from_numpy(np.asarray(im16, dtype=np.uint8).copy())[None].to(engine.dev)
        msk_u8 = torch.from_numpy(np.asarray(mk16, dtype=np.uint8).copy())[None].to(engine.dev)
        # ---- backgrounds of one sample share image, mask and size: they are generated as ONE batch (each with its own
        # seed / generator draws and its own Redux prior), which is ~8 % faster per composite than one at a time
        jobs = []
        for bg_idx, bg_path in enumerate(bgs):
            name = os.path.basename(bg_path)
            suffix = f"_{name.split('rank')[1].split('.')[0]}" if "rank" in name else f"_{bg_idx + 1}"
            mask_path = os.path.join(out_dir, f"{prefix}_mask{suffix}.png")
            mask_img.save(mask_path)
            try:
                bg = H.load_image_rgb(bg_path)
            except Exception as e:
                print(f"加载背景图像 {bg_path} 失败: {str(e)}")
                continue
            bg_saved = os.path.join(out_dir, f"{prefix}_bg{suffix}_original.png")
            shutil.copy(bg_path, bg_saved)
            jobs.append(dict(bg_idx=bg_idx, bg_path=bg_path, name=name, suffix=suffix, mask_path=mask_path, bg=bg, bg_saved=bg_saved,
                             seed=rng.randint(0, 2 ** 32 - 1)))
        for c0 in range(0, len(jobs), max(1, args.bg_batch)):
            chunk = jobs[c0:c0 + max(1, args.bg_batch)]
            n = len(chunk)
            priors = [engine.prior_embeds([jb["bg"]], prompt, [ips], [1.0]) for jb in chunk]
            pe, pp = torch.cat([a for a, _ in priors], 0), torch.cat([b for _, b in priors], 0)
            draws = [generator_noise(jb["seed"], 1, H16, W16, 3) for jb in chunk]      # per image: enc, noise, masked-enc (pipeline order)
            enc_n = torch.cat([d[0] for d in draws], 0)
            noise = torch.cat([pack_noise(d[1]) for d in draws], 0)
            menc_n = torch.cat([d[2] for d in draws], 0)
            outs_dev = engine.pipe(img_u8.expand(n, -1, -1, -1).contiguous(), msk_u8.expand(n, -1, -1).contiguous(), pe, pp,
                                   guidance_scale=guidance, num_inference_steps=args.num_inference_steps, strength=strength,
                                   enc_noise=enc_n.to(engine.dev), masked_enc_noise=menc_n.to(engine.dev),
                                   noise_tokens=noise.to(engine.dev))
            gpu_png = None
            if getattr(args, "png", "host") == "gpu":
                from .. import png as gpu_png
                hires_files = gpu_png.encode(outs_dev)
            outs = outs_dev.cpu().numpy()
            for k_out, (jb, arr) in enumerate(zip(chunk, outs)):
                bg_idx, bg_path, name, suffix, mask_path, bg_saved, seed = (jb[k] for k in ("bg_idx", "bg_path", "name", "suffix", "mask_path", "bg_saved", "seed"))
                result = Image.fromarray(arr)
                hires_path = os.path.join(out_dir, f"{prefix}_hires_result{suffix}.png")
                final = H.downscale_image(result, up) if wu else (H.upscale_image(result, 1.0 / down) if wd else result)
                final_path = os.path.join(out_dir, f"{prefix}_final_result{suffix}.png")
                if gpu_png is not None:
                    with open(hires_path, "wb") as f:
                        f.write(hires_files[k_out])
                    if final is result:
                        final_file = hires_files[k_out]
                    else:                     # the resize back to the original resolution is the host's (PIL); its pixels go up once
                        final_file = gpu_png.encode(torch.from_numpy(np.array(final.convert("RGB"))).to(engine.dev))[0]
                    with open(final_path, "wb") as f:
                        f.write(final_file)
                else:
                    save(result, hires_path)
                    save(final, final_path)
                params = {"categories": cats, "image_scale": 1.0, "prompt_scale": 1.0, "image_prompt_scale": ips,
                          "guidance_scale": guidance, "num_inference_steps": args.num_inference_steps, "strength": strength,
                          "redux_prompt": prompt, "seed": seed, "process_id": process_id, "shot_number": shot, "bg_index": bg_idx,
                          "bg_filename": name, "original_bg_path": bg_path, "copied_bg_path": bg_saved,
                          "original_resolution": {"width": original.width, "height": original.height},
                          "processed_resolution": {"width": processed.width, "height": processed.height},
                          "min_dimension_used": min_dim, "up_scale_factor": up, "down_scale_factor": down, "was_upscaled": wu,
                          "was_downscaled": wd, "bbox_coords_list": bboxes, "processed_bbox_coords_list": pb,
                          "image_id": info["id"] if info.get("id") is not None else "unknown", "num_bbox": len(bboxes)}
                params_path = os.path.join(out_dir, f"{prefix}_params{suffix}.json")
                with open(params_path, "w") as f:
                    json.dump(params, f, indent=2)
                log["outpainted_images"].append({"original_bg_path": bg_path, "copied_bg_path": bg_saved, "hires_result_path": hires_path,
                                                 "final_result_path": final_path, "mask_path": mask_path, "params_path": params_path,
                                                 "bbox_coords_list": bboxes, "processed_bbox_coords_list": pb, "params": params})
        log["original_saved_path"] = orig_saved
        log["status"] = "completed"
    except Exception as e:
        log["status"], log["error"] = "error", str(e)
        print(f"处理样本 {sample_id} 时出错: {str(e)}")
        traceback.print_exc()
    finally:
        dt = time.time() - t0
        log["process_time_seconds"] = dt
        # the resume parser greps exactly these two lines (:1356,:1358)
        print(f"样本 {sample_id} 处理完成，耗时 {dt:.2f} 秒" if log["status"] == "completed" else f"样本 {sample_id} 处理失败，耗时 {dt:.2f} 秒")
    return log


def collect_final_results(process_id, shot, source_process_id=None):
    """copy_final_results_to_collection (:1813-1886).  ``source_process_id``: collect another process's outputs into this
    process's collection (the per-GPU ``<P>_gpu<g>`` trees of a multi-GPU run; the reference scans only ``process_<P>``, which
    then holds nothing but the merged JSON, and collects no images)."""
    root = f"./outpaint_hires/process_{source_process_id or process_id}"
    dest_root = f"./final_results/process_{process_id}/{shot}_shot"
    n = 0
    for ds in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        src = os.path.join(root, ds, f"{shot}_shot")
        if not os.path.isdir(src):
            continue
        dest = os.path.join(dest_root, ds, f"{shot}_shot")
        os.makedirs(dest, exist_ok=True)
        for f in glob.glob(os.path.join(src, "*", "*_final_result*.png")):
            shutil.copy(f, os.path.join(dest, os.path.basename(f)))
            n += 1
    print(f"已收集 {n} 个最终结果到 {dest_root}")
    return dest_root


def run_rank(args, datasets, process_id, rank, world, gpu_process_id=None):
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(torch.cuda.device_count(), 1)   # (more ranks than GPUs: share them)
    torch.cuda.set_device(local)
    engine = Engine("fill", args.model_root, synthetic=args.synthetic_weights, tiny=args.tiny, device=torch.device("cuda", local))
    rng = random.Random(None if args.seed is None else args.seed + rank)
    writer = ImageWriter(args.io_workers)
    done, failed = parse_resume_log(args.log_file) if (args.resume or args.failed_only) else (set(), set())
    outs = {}
    for ds in datasets:
        sdirs = find_sample_dirs(args.result_dir, ds, args.shot)
        ids = sorted(sdirs)
        if args.sample_id:
            ids = [s for s in ids if s == args.sample_id]
        if args.failed_only:
            ids = [s for s in ids if s in failed]
        elif args.resume:
            ids = [s for s in ids if s not in done]
        mine = H.split_samples_for_gpus(ids, world)[rank] if world > 1 else ids
        logs = [process_sample(engine, args, ds, s, sdirs[s], args.shot, gpu_process_id or process_id, rng, writer) for s in mine]
        # the PNGs of the last samples may still be encoding: wait, and turn a failed write into a failed sample
        for path, err in writer.flush():
            print(f"保存图像失败 {path}: {err}")
            for lg in logs:
                if lg["status"] == "completed" and os.path.join(ds, f"{args.shot}_shot", lg["sample_id"]) in path:
                    lg["status"], lg["error"] = "error", f"{path}: {err}"
                    print(f"样本 {lg['sample_id']} 处理失败，耗时 {lg.get('process_time_seconds', 0):.2f} 秒")
        res = H.formatted_result_json(ds, logs, args.shot, gpu_process_id or process_id)
        if gpu_process_id:
            res["gpu_process_id"] = gpu_process_id
        out_dir = os.path.join(f"./outpaint_hires/process_{gpu_process_id or process_id}", ds, f"{args.shot}_shot")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"outpaint_results_{args.shot}shot.json"), "w", encoding="utf-8") as f:
            json.dump(res, f, indent=2, ensure_ascii=False)
        outs[ds] = res
    writer.close()
    return outs


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    process_id = args.process_id or time.strftime("%Y%m%d_%H%M%S")
    if args.custom_upscale:
        try:
            n, d = args.custom_upscale.split(":")
            if n in H.UPSCALE_DIMENSION:
                H.UPSCALE_DIMENSION[n] = int(d)
        except Exception as e:
            print(f"解析自定义上采样维度时出错: {str(e)}")
    if args.collect_only:
        collect_final_results(process_id, args.shot)
        return 0
    datasets = [args.dataset] if args.dataset else list(H.STRENGTH)        # every key of strength_params is accepted
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.multi_gpu and world == 1 and "DRAG_CHILD" not in os.environ:
        n = args.num_gpus or torch.cuda.device_count()
        procs = []
        for g in range(n):
            env = dict(os.environ, DRAG_CHILD="1", RANK=str(g), WORLD_SIZE=str(n), LOCAL_RANK=str(g))
            procs.append(subprocess.Popen([sys.executable, "-m", "domain_rag_amd.cli.stage3_outpaint"] + argv + ["--process_id", process_id], env=env))
        rc = [p.wait() for p in procs]
        for ds in datasets:
            parts = []
            for g in range(n):
                pth = os.path.join(f"./outpaint_hires/process_{H.create_gpu_process_id(process_id, g)}", ds, f"{args.shot}_shot",
                                   f"outpaint_results_{args.shot}shot.json")
                if os.path.exists(pth):
                    parts.append(json.load(open(pth, encoding="utf-8")))
            if parts:
                merged = H.merge_gpu_results(ds, parts, args.shot, process_id)
                od = os.path.join(f"./outpaint_hires/process_{process_id}", ds, f"{args.shot}_shot")
                os.makedirs(od, exist_ok=True)
                with open(os.path.join(od, f"outpaint_results_{args.shot}shot.json"), "w", encoding="utf-8") as f:
                    json.dump(merged, f, indent=2, ensure_ascii=False)
        for g in range(n):
            collect_final_results(process_id, args.shot, source_process_id=H.create_gpu_process_id(process_id, g))
        return max(rc) if rc else 0
    gpid = H.create_gpu_process_id(process_id, rank) if world > 1 else None
    run_rank(args, datasets, process_id, rank, world, gpid)
    if world == 1:
        collect_final_results(process_id, args.shot)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
