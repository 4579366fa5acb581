"""The stage CLIs (LaMa -> retrieval -> generation -> outpainting) end to end on a synthetic mini-dataset (tiny architectures, seeded synthetic weights):
file tree + JSON schemas of the reference's stage boundaries."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mod, args, cwd, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT, DRAG_TIMESTAMP="20260101_000000", **(env_extra or {}))
    r = subprocess.run([sys.executable, "-m", mod] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout + (r.stderr if mod.endswith("stage0_lama") else "")        # stage 0 logs through `logging` (stderr)


def test_three_stages_on_mini_dataset(gpu, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    root = tmp_path
    (root / "retrieval" / "coco" / "train2017").mkdir(parents=True)
    for i in range(8):
        Image.fromarray(rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)).save(root / "retrieval" / "coco" / "train2017" / f"{i:06d}.jpg")
    ds = "ArTaxOr"
    (root / "lama_inpaint").mkdir()
    (root / "datasets" / ds / "annotations").mkdir(parents=True)
    (root / "datasets" / ds / "train").mkdir(parents=True)
    images, anns = [], []
    for i, name in enumerate(["beetle_01", "moth_02"]):
        arr = rng.integers(0, 256, (48, 72, 3), dtype=np.uint8)
        Image.fromarray(arr).save(root / "datasets" / ds / "train" / f"{name}.jpg")
        images.append({"id": i + 1, "file_name": f"{name}.jpg", "width": 72, "height": 48})
        anns.append({"id": i + 1, "image_id": i + 1, "bbox": [10 + i, 8, 20, 16], "category_id": 1})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "Coleoptera"}]},
              open(root / "datasets" / ds / "annotations" / "1_shot.json", "w"))

    # ---- stage 0 (run from ./lama_inpaint like inapint.sh): the object-free k-shot images stages 1 and 2 read
    out0 = _run("domain_rag_amd.cli.stage0_lama", ["--datasets", ds, "--shots", "1", "--synthetic-weights", "--tiny"], cwd=root / "lama_inpaint")
    assert "完成 2 个图像, 失败 0 个" in out0
    for name in ("beetle_01", "moth_02"):
        assert Image.open(root / "lamainpaint" / ds / "1_shot" / f"{name}.jpg").size == (72, 48)

    # ---- stage 1 (run from ./retrieval like domainrag.sh)
    _run("domain_rag_amd.cli.stage1_retrieval", ["--datasets", ds, "--shots", "1", "--coco-dir", "./coco", "--clip-top-k", "6",
                                                 "--pretrained-coco-features", "none.pt", "--synthetic-weights"], cwd=root / "retrieval")
    rr = root / "retrieval" / "retrieval_results"
    allr = json.load(open(rr / "all_shots_retrieval_results.json"))
    entry = allr[ds]["1_shot"]["beetle_01"][0]
    assert entry["sample_id"] == "beetle_01" and entry["category"] == "beetle_01"      # category == sample id
    sims = entry["similar_images"]
    assert len(sims) == 6 and [s["rank"] for s in sims] == [1, 2, 3, 4, 5, 6]
    assert set(sims[0]) == {"rank", "similarity", "image_path", "source_dataset"} and sims[0]["source_dataset"] == "coco"
    assert np.load(rr / "coco_clip_features.npy").shape == (8, 512) and len(json.load(open(rr / "coco_image_paths.json"))) == 8
    assert (rr / f"{ds}_1_shot_beetle_01_beetle_01_retrieval_results.json").exists()
    assert Image.open(rr / f"{ds}_1_shot_beetle_01_beetle_01_visual.jpg").size == (1200, 729)
    assert np.load(rr / f"{ds}_1_shot_inpainted_clip_features.npy").shape == (2, 512)

    # ---- stage 2
    _run("domain_rag_amd.cli.stage2_generate", ["--dataset", ds, "--shots", "1", "--retrieval_results_dir", str(rr), "--output_dir", "result",
                                                "--coco_dir", "./retrieval/coco/train2017", "--synthetic-weights", "--tiny",
                                                "--num_inference_steps", "2"], cwd=root)
    base = root / "result" / f"{ds}_1shot_retrieval" / "results_coco_0.8_target_1.0_cocotext_1.0_targettext_1.0_20260101_000000"
    for r in range(1, 6):
        im = Image.open(base / "beetle_01" / f"generated_image_rank{r}.png")
        assert im.size == (64, 64)
        assert (base / "beetle_01" / f"ref_inputrank{r}.jpg").exists()
    assert (base / "beetle_01" / "target_input.png").exists() and (base / "beetle_01" / "params.txt").exists() and (base / "batch_params.txt").exists()
    bp = open(base / "batch_params.txt", encoding="utf-8").read()
    assert "处理样本数: 2" in bp and "成功处理样本数: 2" in bp and "失败处理样本数: 0" in bp and "总共生成图像数: 10" in bp
    assert not (base / "beetle_01" / "error.txt").exists() and not (base / "beetle_01" / "generation_failed.txt").exists()

    # ---- stage 3
    out = _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "7", "--dataset", ds, "--shot", "1", "--synthetic-weights", "--tiny",
                                                       "--num_inference_steps", "2", "--seed", "3"], cwd=root)
    assert "样本 beetle_01 处理完成" in out and "样本 moth_02 处理完成" in out
    sd = root / "outpaint_hires" / "process_7" / ds / "1_shot" / "beetle_01"
    pre = f"{ds}_beetle_01_1shot"
    for r in range(1, 6):
        assert Image.open(sd / f"{pre}_final_result_{r}.png").size == (72, 48)      # back at the original resolution
        hires = Image.open(sd / f"{pre}_hires_result_{r}.png").size
        assert hires == (96, 64)                                                     # min side 48 -> 64, x16 grid
        prm = json.load(open(sd / f"{pre}_params_{r}.json"))
        assert prm["strength"] == 0.9 and prm["guidance_scale"] == 30.0 and prm["was_upscaled"] and prm["num_bbox"] == 1
        assert prm["processed_bbox_coords_list"] == [[int(c * prm["up_scale_factor"]) for c in [10, 8, 20, 16]]]
        assert (sd / f"{pre}_bg_{r}_original.png").exists()
        m = np.asarray(Image.open(sd / f"{pre}_mask_{r}.png"))
        assert m.shape == (64, 96) and set(np.unique(m)) == {0, 255}
    assert Image.open(sd / f"{pre}_bbox1_original.jpg").size == (20, 16) and Image.open(sd / f"{pre}_upscaled_bg.png").size == (96, 64)
    assert (sd / f"{pre}_original.png").exists() and not (sd / f"{pre}_downscaled_bg.png").exists()
    res = json.load(open(root / "outpaint_hires" / "process_7" / ds / "1_shot" / "outpaint_results_1shot.json"))
    assert res["dataset"] == ds and res["shot_number"] == 1 and len(res["samples"]) == 2
    assert len(res["samples"][0]["outpainted_images"]) == 5 and res["samples"][0]["categories"] == ["Coleoptera"]
    assert res["samples"][0]["bbox_saved_paths"][0].endswith(f"{pre}_bbox1_original.jpg") and res["samples"][0]["bbox_image_sizes"] == [[20, 16]]
    fin = root / "final_results" / "process_7" / "1_shot" / ds / "1_shot"
    assert len(list(fin.glob("*_final_result*.png"))) == 10
    # the backgrounds of a sample are composited as one batch; one at a time (like the reference) must give the same pixels
    _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "8", "--dataset", ds, "--shot", "1", "--synthetic-weights", "--tiny",
                                                 "--num_inference_steps", "2", "--seed", "3", "--bg_batch", "1"], cwd=root)
    sd8 = root / "outpaint_hires" / "process_8" / ds / "1_shot" / "beetle_01"
    for r in range(1, 6):
        assert np.array_equal(np.asarray(Image.open(sd / f"{pre}_hires_result_{r}.png")), np.asarray(Image.open(sd8 / f"{pre}_hires_result_{r}.png")))
        assert json.load(open(sd / f"{pre}_params_{r}.json"))["seed"] == json.load(open(sd8 / f"{pre}_params_{r}.json"))["seed"]
    # ---- stage 3, --multi_gpu with two worker processes (they share the one GPU of the test box): contiguous sharding,
    # per-GPU process ids, merged manifest with the multi-GPU bookkeeping, images collected from both workers
    _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "9", "--dataset", ds, "--shot", "1", "--synthetic-weights", "--tiny",
                                                 "--num_inference_steps", "2", "--seed", "3", "--multi_gpu", "--num_gpus", "2"], cwd=root)
    merged = json.load(open(root / "outpaint_hires" / "process_9" / ds / "1_shot" / "outpaint_results_1shot.json"))
    assert merged["multi_gpu"] is True and merged["num_gpus"] == 2 and merged["gpu_process_ids"] == ["9_gpu0", "9_gpu1"]
    assert [smp["sample_id"] for smp in merged["samples"]] == ["beetle_01", "moth_02"]
    assert (root / "outpaint_hires" / "process_9_gpu0" / ds / "1_shot" / "beetle_01").is_dir()
    assert (root / "outpaint_hires" / "process_9_gpu1" / ds / "1_shot" / "moth_02").is_dir()
    assert len(list((root / "final_results" / "process_9" / "1_shot" / ds / "1_shot").glob("*_final_result*.png"))) == 10
    # worker 0 draws its seeds from args.seed + 0 like the single-process run: same pixels for its samples
    sd9 = root / "outpaint_hires" / "process_9_gpu0" / ds / "1_shot" / "beetle_01"
    assert np.array_equal(np.asarray(Image.open(sd / f"{pre}_hires_result_1.png")), np.asarray(Image.open(sd9 / f"{pre}_hires_result_1.png")))


def test_stage3_background_writer_matches_inline_files(gpu, tmp_path):
    """--io_workers 0 (Image.save on the generation thread, the reference's pattern) and the default background encoder
    processes must leave byte-identical PNGs"""
    from PIL import Image
    rng = np.random.default_rng(1)
    root = tmp_path
    ds, name = "DIOR", "plane_7"
    (root / "datasets" / ds / "annotations").mkdir(parents=True); (root / "datasets" / ds / "train").mkdir(parents=True)
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(root / "datasets" / ds / "train" / f"{name}.jpg")
    json.dump({"images": [{"id": 1, "file_name": f"{name}.jpg", "width": 56, "height": 40}],
               "annotations": [{"id": 1, "image_id": 1, "bbox": [8, 6, 20, 14], "category_id": 1}], "categories": [{"id": 1, "name": "airplane"}]},
              open(root / "datasets" / ds / "annotations" / "1_shot.json", "w"))
    sdir = root / "result" / f"{ds}_1shot_retrieval" / "results_x" / name
    sdir.mkdir(parents=True)
    for r in (1, 2):
        Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(sdir / f"generated_image_rank{r}.png")
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)).save(sdir / "target_input.png")
    common = ["--dataset", ds, "--shot", "1", "--synthetic-weights", "--tiny", "--num_inference_steps", "2", "--seed", "5"]
    _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "a"] + common + ["--io_workers", "0"], cwd=root)
    _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "b"] + common, cwd=root)
    da, db = (root / "outpaint_hires" / f"process_{p}" / ds / "1_shot" / name for p in "ab")
    pngs = sorted(f for f in os.listdir(da) if f.endswith(".png"))
    assert len(pngs) >= 8 and pngs == sorted(f for f in os.listdir(db) if f.endswith(".png"))
    for f in pngs:
        assert (da / f).read_bytes() == (db / f).read_bytes(), f
    assert len(list((root / "final_results" / "process_b" / "1_shot" / ds / "1_shot").glob("*_final_result*.png"))) == 2
    # --png host (Pillow's zlib, the reference's encoder) vs the default device encoder: other bytes for the hires / final
    # results, the same pixels in every file
    _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "c"] + common + ["--png", "host"], cwd=root)
    dc = root / "outpaint_hires" / "process_c" / ds / "1_shot" / name
    assert pngs == sorted(f for f in os.listdir(dc) if f.endswith(".png"))
    differ = 0
    for f in pngs:
        a, c = Image.open(da / f), Image.open(dc / f)
        assert a.mode == c.mode and a.size == c.size and np.array_equal(np.asarray(a), np.asarray(c)), f
        differ += (da / f).read_bytes() != (dc / f).read_bytes()
    assert differ == 4          # 2 x (hires, final)


def test_stage3_downscale_branch(gpu, tmp_path):
    """an original wider than the 2800-px cap: processed at 2800 wide, composited, scaled back up (outpainting_…:403-458,480-498);
    the params / manifest record the down-scale and no *_upscaled_bg.png is written"""
    from PIL import Image
    rng = np.random.default_rng(2)
    root = tmp_path
    ds, name = "DIOR", "wide_1"
    (root / "datasets" / ds / "annotations").mkdir(parents=True); (root / "datasets" / ds / "train").mkdir(parents=True)
    W, H = 2900, 420
    Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(root / "datasets" / ds / "train" / f"{name}.jpg")
    json.dump({"images": [{"id": 1, "file_name": f"{name}.jpg", "width": W, "height": H}],
               "annotations": [{"id": 1, "image_id": 1, "bbox": [1000, 100, 600, 200], "category_id": 1}], "categories": [{"id": 1, "name": "ship"}]},
              open(root / "datasets" / ds / "annotations" / "1_shot.json", "w"))
    sdir = root / "result" / f"{ds}_1shot_retrieval" / "results_x" / name
    sdir.mkdir(parents=True)
    Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(sdir / "generated_image_rank1.png")
    out = _run("domain_rag_amd.cli.stage3_outpaint", ["--process_id", "d", "--dataset", ds, "--shot", "1", "--synthetic-weights", "--tiny",
                                                       "--num_inference_steps", "2", "--seed", "5"], cwd=root)
    assert f"样本 {name} 处理完成" in out
    sd = root / "outpaint_hires" / "process_d" / ds / "1_shot" / name
    pre = f"{ds}_{name}_1shot"
    prm = json.load(open(sd / f"{pre}_params_1.json"))
    assert prm["was_downscaled"] and not prm["was_upscaled"] and abs(prm["down_scale_factor"] - 2800 / 2900) < 1e-12
    assert prm["processed_resolution"] == {"width": 2800, "height": int(420 * (2800 / 2900))}
    assert prm["processed_bbox_coords_list"] == [[int(v * prm["down_scale_factor"]) for v in [1000, 100, 600, 200]]]
    assert Image.open(sd / f"{pre}_downscaled_bg.png").size == (2800, 405) and not (sd / f"{pre}_upscaled_bg.png").exists()
    hires = Image.open(sd / f"{pre}_hires_result_1.png").size
    assert hires == (2800, 400)                                                   # multiples of 16 inside the Fill pipeline
    fw, fh = Image.open(sd / f"{pre}_final_result_1.png").size                      # scaled back by 1 / down_scale_factor
    assert abs(fw - 2900) <= 1 and abs(fh - int(400 / prm["down_scale_factor"])) <= 1
