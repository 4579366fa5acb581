"""Host logic (resolution policy, masks, sharding, manifests, path cleaning) pinned against goldens captured from
the IMPORTED reference (tests/golden/host_logic.json <- tests/golden/make_host_goldens.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_logic.json")))


def test_resolution_policy():
    from PIL import Image
    from domain_rag_amd import hostlogic as H
    for c in GOLD["process_image_resolution"]:
        if c.get("error"):
            with pytest.raises(ValueError):
                H.resolution_plan(c["w"], c["h"], c["min"], 2800)
            continue
        nw, nh, up, down, wu, wd = H.resolution_plan(c["w"], c["h"], c["min"], 2800)
        assert [nw, nh] == c["out"] and up == c["up"] and down == c["down"] and wu == c["wu"] and wd == c["wd"], c
        img, up2, down2, wu2, wd2 = H.process_image_resolution(Image.new("RGB", (c["w"], c["h"])), c["min"], 2800)
        assert list(img.size) == c["out"] and (up2, down2, wu2, wd2) == (up, down, wu, wd)
    for c in GOLD["downscale_image"]:
        assert list(H.downscale_image(Image.new("RGB", (c["w"], c["h"])), c["s"]).size) == c["out"]
    # SURVEY-verified identities
    assert H.resolution_plan(500, 375)[:2] == (1365, 1024) and H.resolution_plan(4000, 3000)[:2] == (2800, 2100)


def test_outpaint_mask_bitmaps():
    from PIL import Image
    from domain_rag_amd import hostlogic as H
    for c in GOLD["generate_outpaint_mask"]:
        a = H.outpaint_mask_array(c["W"], c["H"], c["bboxes"])
        assert a.shape == (c["H"], c["W"]) and int((a == 255).sum()) == c["white"], c
        assert hashlib.sha256(a.tobytes()).hexdigest() == c["sha256"], c
        m, bbs = H.generate_outpaint_mask(Image.new("RGB", (c["W"], c["H"])), c["bboxes"])
        assert m.mode == c["mode"] and np.array_equal(np.asarray(m), a) and bbs == c["bboxes"]
    a = H.outpaint_mask_array(64, 48, [[10, 10, 20, 15]])       # inclusive rectangle: 21 x 16 px kept
    assert abs((a == 255).mean() - 0.890625) < 1e-12


def test_sharding_and_manifests():
    from domain_rag_amd import hostlogic as H
    from domain_rag_amd.retrieval import shard_bounds
    for c in GOLD["split_samples_for_gpus"]:
        chunks = H.split_samples_for_gpus(list(range(c["n"])), c["g"])
        assert [len(x) for x in chunks] == c["sizes"] and [x[0] if x else None for x in chunks] == c["first"]
        if c["g"] > 1:   # the rank-local form used by the data-parallel drivers agrees with it
            assert [list(range(*shard_bounds(c["n"], c["g"], r))) for r in range(c["g"])] == chunks
    assert H.split_samples_for_gpus(list(range(10)), 4) == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    gj = GOLD["generate_formatted_result_json"]
    out = H.formatted_result_json("DS", gj["logs"], gj["shot"], process_id="golden")
    out.pop("timestamp")
    assert out == gj["out"]
    mg = GOLD["merge_gpu_results"]
    assert H.merge_gpu_results("DS", mg["in"], 1) == mg["out"]
    assert H.create_gpu_process_id("7", 3) == GOLD["create_gpu_process_id"]


def test_tables_and_paths():
    from domain_rag_amd import hostlogic as H
    from domain_rag_amd.retrieval import clean_image_path
    t = GOLD["tables"]
    assert H.STRENGTH == t["strength"] and H.GUIDANCE_SCALE == t["guidance"] and H.IMAGE_PROMPT_SCALE == t["image_prompt_scale"]
    assert H.UPSCALE_DIMENSION == t["upscale"] and H.REDUX_PROMPT == t["redux_prompt"]
    assert (H.DEFAULT_STRENGTH, H.DEFAULT_GUIDANCE, H.MIN_DIMENSION, H.MAX_DIMENSION) == (
        t["default_strength"], t["default_guidance"], t["min_dim"], t["max_dim"])
    for c in GOLD["clean_image_path"]:
        assert clean_image_path(c["in"]) == c["out"]


def test_bbox_helpers():
    from domain_rag_amd import hostlogic as H
    assert H.scale_bboxes([[10, 20, 30, 40]], 2.7306666, 1.0, True, False) == [[27, 54, 81, 109]]
    assert H.scale_bboxes([[10, 20, 30, 40]], 1.0, 0.7, False, True) == [[7, 14, 21, 28]]
    assert H.scale_bboxes([[1.5, 2, 3, 4]], 1.0, 1.0, False, False) == [[1.5, 2, 3, 4]]
    ann = {"images": [{"id": 7, "file_name": "abc_001.jpg"}, {"id": "8", "file_name": "zzz.png"}],
           "annotations": [{"image_id": "7", "bbox": [1, 2, 3, 4], "category_id": 2}, {"image_id": 7, "bbox": [5, 6, 7, 8], "category_id": 9},
                           {"image_id": 8, "bbox": [0, 0, 1, 1], "category_id": 2}],
           "categories": [{"id": 2, "name": "beetle"}]}
    info, boxes, cats = H.lookup_sample_annotations(ann, "abc_001")
    assert info["id"] == 7 and boxes == [[1, 2, 3, 4], [5, 6, 7, 8]] and cats == ["beetle", "unknown"]
    assert H.lookup_sample_annotations(ann, "abc")[0]["id"] == 7          # substring fallback
    assert H.lookup_sample_annotations(ann, "nope") is None
    assert H.crop_box([-3, 5.9, 1000, "2"], 100, 50) == (0, 5, 100, 7)


def test_oracle_stem_identity():
    """calc_mean_std of arange(96).reshape(2,3,4,4) — the value the reference itself produces"""
    import torch
    from oracle import stem
    m, s = stem.calc_mean_std(torch.arange(96.0).reshape(2, 3, 4, 4))
    assert np.allclose(m.flatten().numpy(), GOLD["calc_mean_std"]["mean"]) and np.allclose(s.flatten().numpy(), GOLD["calc_mean_std"]["std"])


def test_annotation_lookup_matches_reference():
    """get_bbox_and_original_image (outpainting_…:570-682) on a synthetic COCO-format dataset: matched image, all its
    boxes (raw values), categories, and the clamped crop sizes — goldens captured from the imported reference"""
    from domain_rag_amd import hostlogic as H
    gg = GOLD["get_bbox_and_original_image"]
    ann, files = gg["annotations"], gg["files"]
    for c in gg["cases"]:
        found = H.lookup_sample_annotations(ann, c["sample_id"])
        on_disk = found is not None and found[0]["file_name"] in files            # the reference also needs the file to exist
        assert on_disk == c["found"], c
        if not on_disk:
            continue
        info, boxes, cats = found
        w, h = files[info["file_name"]]
        assert [w, h] == c["image_size"] and boxes == c["bboxes"] and cats == c["categories"] and info["id"] == c["image_id"]
        crops = [H.crop_box(b, w, h) for b in boxes]
        assert [[x1 - x0, y1 - y0] for x0, y0, x1, y1 in crops] == c["crop_sizes"], c


def test_second_stage_rerank_matches_reference():
    """resnet_second_stage_rerank (retrieval/…:454-497) with injected style vectors: L2 distance in fp32, stable sort
    (a tie keeps CLIP order), unreadable candidates dropped, similarity 1/(1+d), cleaned paths, source default"""
    from domain_rag_amd import retrieval as R
    gg = GOLD["resnet_second_stage_rerank"]
    vecs = {k: (np.asarray(v, dtype=np.float32) if v is not None else None) for k, v in gg["style"].items()}

    class FakeStem:
        def features_from_path(self, path):
            return vecs.get(path)
    out = R.resnet_second_stage_rerank(gg["query"], gg["first"], FakeStem())
    assert out == gg["out"]
    cache: dict = {}
    assert R.resnet_second_stage_rerank(gg["query"], gg["first"], FakeStem(), cache) == gg["out"] and len(cache) == 11
    assert R.resnet_second_stage_rerank("unreadable.jpg", gg["first"][:3], FakeStem()) == gg["out_query_unreadable"]


def test_randomised_sweeps_match_reference():
    """300 random sizes through the resolution policy and 120 random (float / negative / off-canvas) box sets through the
    outpaint mask, expected values captured from the imported reference"""
    import hashlib
    from domain_rag_amd import hostlogic as H
    for row in GOLD["process_image_resolution_sweep"]:
        w, h, mind = row[:3]
        if row[3] is None:
            with pytest.raises(ValueError):
                H.resolution_plan(w, h, mind, 2800)
        else:
            assert list(H.resolution_plan(w, h, mind, 2800)) == row[3:], row
    for c in GOLD["generate_outpaint_mask_sweep"]:
        if "error" in c:
            with pytest.raises(Exception):
                H.outpaint_mask_array(c["W"], c["H"], c["bboxes"])
            continue
        m = H.outpaint_mask_array(c["W"], c["H"], c["bboxes"])
        assert int((m == 255).sum()) == c["white"] and hashlib.sha256(m.tobytes()).hexdigest() == c["sha256"], c


def test_cv2_style_linear_resize_properties():
    """the OpenCV INTER_LINEAR restatement used for the stem input (cv2 itself is not importable: unpinned): identity at equal
    size, within 1 LSB of exact two-tap half-pixel bilinear interpolation, constant images stay constant, and — unlike
    PIL's BILINEAR — no low-pass when shrinking (white noise keeps about twice the contrast)"""
    from PIL import Image
    from domain_rag_amd.retrieval import cv2_resize_linear_u8
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)
    assert np.array_equal(cv2_resize_linear_u8(img, 90, 60), img)
    assert np.all(cv2_resize_linear_u8(np.full((33, 47, 3), 201, np.uint8), 256, 256) == 201)
    for (ow, oh) in [(256, 256), (45, 30), (200, 17), (90, 200)]:
        h, w = img.shape[:2]
        fx = (np.arange(ow) + 0.5) * w / ow - 0.5; fy = (np.arange(oh) + 0.5) * h / oh - 0.5
        x0 = np.floor(fx).astype(int); ax = fx - x0; ax[(x0 < 0) | (x0 >= w - 1)] = 0; x0 = np.clip(x0, 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1)
        y0f = np.floor(fy).astype(int); ay = fy - y0f; y0 = np.clip(y0f, 0, h - 1); y1 = np.clip(y0f + 1, 0, h - 1)
        f = img.astype(np.float64)
        r = f[:, x0] * (1 - ax)[None, :, None] + f[:, x1] * ax[None, :, None]
        ref = r[y0] * (1 - ay)[:, None, None] + r[y1] * ay[:, None, None]
        assert np.abs(cv2_resize_linear_u8(img, ow, oh).astype(np.float64) - ref).max() <= 1.0
    noise = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)            # white noise: a 2-tap resize keeps ~2x PIL's contrast
    ours = cv2_resize_linear_u8(noise, 256, 256)
    pil = np.asarray(Image.fromarray(noise).resize((256, 256), Image.BILINEAR))
    assert ours.std() > 40 and pil.std() < 30


def test_timestep_rounding_chain():
    """the value the DiT finally embeds is bf16(bf16(bf16(t) / 1000) * 1000) — t is cast to the latents' dtype BEFORE the
    pipeline divides by 1000, and the transformer multiplies back in bf16.  ``scheduler.model_timestep`` + the
    transformer's own ``to(bf16) * 1000`` must reproduce that chain for every timestep of the schedules in use."""
    import torch
    from domain_rag_amd.scheduler import flow_sigmas, model_timestep
    n_diff = 0
    for n in (4, 20, 30, 50):
        for seq in (1024, 4096, 5440, 16384):
            for t in flow_sigmas(n, seq)[1]:
                t32 = torch.tensor(float(t), dtype=torch.float32)
                ref = (t32.to(torch.bfloat16) / 1000).to(torch.bfloat16) * 1000            # pipeline, then transformer
                ours = torch.tensor(model_timestep(float(t))).to(torch.bfloat16) * 1000   # what FluxTransformerHIP._set_times does
                assert float(ref) == float(ours)
                n_diff += float(ref) != float((t32 / 1000).to(torch.bfloat16) * 1000)     # the naive chain is NOT the same
    assert n_diff > 50


def test_load_image_applies_exif_orientation(tmp_path):
    """stages 2 / 3 open images with diffusers' load_image, which transposes by the EXIF orientation tag"""
    from PIL import Image
    from domain_rag_amd import hostlogic as H
    arr = np.zeros((20, 30, 3), np.uint8); arr[:5] = 255            # bright top edge, 30 wide x 20 high
    im = Image.fromarray(arr)
    ex = im.getexif(); ex[0x0112] = 6                               # "rotate 90 CW to display"
    im.save(tmp_path / "r.jpg", exif=ex, quality=95)
    plain = np.asarray(Image.open(tmp_path / "r.jpg").convert("RGB"))
    fixed = np.asarray(H.load_image_rgb(str(tmp_path / "r.jpg")))
    assert plain.shape == (20, 30, 3) and fixed.shape == (30, 20, 3)
    assert fixed[:, -5:].mean() > 200 and fixed[:, :10].mean() < 50      # the bright edge is now on the right


def test_stem_restatement_matches_upstream_resnet_embeddings():
    """torchvision is not installable here, but `transformers` ships the same ResNet-50 stem (microsoft/resnet-50 is a port of
    the torchvision weights): Conv 7x7/2 pad 3 no bias -> BatchNorm(eps 1e-5) -> ReLU -> MaxPool 3x3/2 pad 1.  The oracle's
    functional restatement must equal that upstream module on shared random weights, bit for bit."""
    import torch
    from transformers import ResNetConfig
    from transformers.models.resnet.modeling_resnet import ResNetEmbeddings
    from oracle import stem as ostem
    g = torch.Generator().manual_seed(0)
    st = {"conv1.weight": torch.randn(64, 3, 7, 7, generator=g) * 0.1, "bn1.weight": torch.rand(64, generator=g) + 0.5,
          "bn1.bias": torch.randn(64, generator=g) * 0.1, "bn1.running_mean": torch.randn(64, generator=g) * 0.1,
          "bn1.running_var": torch.rand(64, generator=g) + 0.5}
    m = ResNetEmbeddings(ResNetConfig()).eval()
    conv, bn = m.embedder.convolution, m.embedder.normalization
    assert (conv.kernel_size, conv.stride, conv.padding, conv.bias, bn.eps) == ((7, 7), (2, 2), (3, 3), None, 1e-5)
    with torch.no_grad():
        conv.weight.copy_(st["conv1.weight"]); bn.weight.copy_(st["bn1.weight"]); bn.bias.copy_(st["bn1.bias"])
        bn.running_mean.copy_(st["bn1.running_mean"]); bn.running_var.copy_(st["bn1.running_var"])
        for shape in ((1, 3, 256, 256), (2, 3, 97, 131)):
            x = torch.rand(shape, generator=g)
            assert torch.equal(m(x), ostem.stem(x, st))
