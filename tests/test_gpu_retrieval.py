"""Stage-1 retrieval objects on the HIP path: CLIP encode_image, resident IndexFlatIP, stem style, two-stage flow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stem_style_vs_oracle(gpu):
    from domain_rag_amd.retrieval import StemStyle
    from oracle import stem as ostem
    st = StemStyle(device=gpu, seed=3)
    x = torch.rand(3, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    got = st(x).cpu()
    ref = ostem.style_vector(x, st.state)
    assert torch.allclose(got, ref, rtol=2e-4, atol=2e-5), (got - ref).abs().max()
    # the verified reference identity: arange(96).reshape(2,3,4,4) -> mean 7.5/23.5/39.5, std 4.7610
    m, s = ostem.calc_mean_std(torch.arange(96.0).reshape(2, 3, 4, 4))
    assert torch.allclose(m[0], torch.tensor([7.5, 23.5, 39.5])) and abs(s[0, 0].item() - 4.7610) < 1e-3
    x2 = torch.rand(1, 3, 96, 64, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(st(x2).cpu(), ostem.style_vector(x2, st.state), rtol=2e-4, atol=2e-5)


def test_clip_full_size_and_index(gpu):
    """real ViT-B/32 shape: float-preprocess path == uint8 path; embeddings vs transformers oracle; top-k on them"""
    from domain_rag_amd import retrieval as R
    from domain_rag_amd.vit import VitConfig, init_generic_params, VitHIP
    from oracle import retrieval as oret, vit as ov
    cfg = VitConfig.clip_vit_b32()
    g = init_generic_params(cfg, 5)
    model = R.ClipImageModel(VitHIP(cfg, g, gpu))
    img = (torch.rand(6, 224, 224, 3, generator=torch.Generator().manual_seed(2)) * 255).to(torch.uint8)
    px = ov.normalize_u8(img, cfg.mean, cfg.std)
    e_u8 = model.encode_image(img).cpu()
    e_f = model.encode_image(px).cpu()
    assert torch.allclose(e_u8, e_f, atol=2e-2 * e_u8.abs().max().item())
    ref = ov.clip_image_embeds(g, 224, 32, 768, 12, 12, 3072, 512, px, torch.float32)
    rel = ((e_u8 - ref).abs().max() / ref.abs().max()).item()
    assert rel < 3e-2, rel
    emb = model.embed_normalized(img).cpu()
    assert torch.allclose(emb.norm(dim=-1), torch.ones(6), atol=1e-5)
    refn = ref / ref.norm(dim=-1, keepdim=True)
    assert (emb * refn).sum(-1).min().item() > 0.999      # cosine between HIP and oracle embeddings
    # resident index: add in two chunks, search == oracle on the same features, bit-exact
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((3000, 512)).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    idx = R.IndexFlatIP(512, gpu)
    idx.add(feats[:1000]); idx.add(feats[1000:])
    assert idx.ntotal == 3000
    D, I = idx.search(feats[:3] + 0.01, 100)
    Dr, Ir = oret.cosine_topk(feats, feats[:3] + 0.01, 100)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)


def test_two_stage_flow_on_files(gpu, tmp_path):
    from PIL import Image
    from domain_rag_amd import retrieval as R
    rng = np.random.default_rng(1)
    paths = []
    for i in range(12):
        p = tmp_path / f"img{i:02d}.jpg"
        Image.fromarray(rng.integers(0, 256, (80, 100, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    model, pre = R.load_clip("ViT-B/32", gpu, seed=1)
    feats, valid = R.compute_corpus_features(model, pre, paths + [str(tmp_path / "missing.jpg")], batch=5)
    assert valid == paths and feats.shape == (12, 512) and feats.dtype == np.float32
    q = feats[4]
    first = R.clip_first_stage_retrieval(q, {"coco": feats}, {"coco": valid}, top_k=100, device=gpu)
    assert len(first) == 12 and first[0]["index"] == 4 and first[0]["source_dataset"] == "coco"
    assert all(first[i]["similarity"] >= first[i + 1]["similarity"] for i in range(11))
    stem = R.StemStyle(device=gpu)
    final = R.resnet_second_stage_rerank(paths[4], first, stem, style_cache={})
    assert [r["rank"] for r in final] == list(range(1, 13))
    assert final[0]["image_path"] == paths[4] and abs(final[0]["similarity"] - 1.0) < 1e-6
    assert set(final[0]) == {"rank", "similarity", "image_path", "source_dataset"}


def test_first_stage_plumbing_matches_reference(gpu):
    """clip_first_stage_retrieval (retrieval/…:396-451): several datasets stacked in dict order, empty / None ones
    skipped, k = min(top_k, N), {similarity, image_path, source_dataset, index} — golden captured from the reference
    (its faiss replaced by an exact numpy inner product; scores well separated, so order is unambiguous)"""
    import json, os
    import numpy as np
    from domain_rag_amd import retrieval as R
    gg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_logic.json")))["clip_first_stage_retrieval"]
    feats = {k: (np.asarray(v, dtype=np.float32).reshape(-1, 64) if v is not None else None) for k, v in gg["features"].items()}
    q = np.asarray(gg["query"], dtype=np.float32)
    assert R.clip_first_stage_retrieval(q, feats, gg["paths"], top_k=100, device=gpu) == gg["out_k100"]
    assert R.clip_first_stage_retrieval(q, feats, gg["paths"], top_k=3, device=gpu) == gg["out_k3"]
    assert R.clip_first_stage_retrieval(q, {"none": None}, {"none": None}, top_k=3, device=gpu) == gg["out_nothing"]


def test_corpus_features_same_bits_through_decode_processes(gpu, tmp_path):
    """thread-pool decode + GPU resize, thread-pool decode + host preprocess, and worker-process decode + PIL resize must give
    the same embeddings, skip the same unreadable files and keep the path order"""
    import numpy as np
    from PIL import Image
    from domain_rag_amd import retrieval as R
    rng = np.random.default_rng(4)
    paths = []
    for i, (h, w) in enumerate([(480, 640), (640, 480), (224, 224), (225, 300), (97, 61), (333, 777)] * 4):
        p = tmp_path / f"{i:03d}.jpg"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p, quality=92)
        paths.append(str(p))
    (tmp_path / "bad.jpg").write_bytes(b"junk")
    paths.insert(7, str(tmp_path / "bad.jpg"))
    model, host_pre = R.load_clip("ViT-B/32", device=gpu)
    a, va = R.compute_corpus_features(model, R.load_clip_device_preprocess(gpu), paths, batch=16, decode_workers=4)
    b, vb = R.compute_corpus_features(model, host_pre, paths, batch=5, decode_workers=2)
    c, vc = R.compute_corpus_features(model, None, paths, batch=7, decode_procs=3)
    assert va == vb == vc == [p for p in paths if "bad" not in p] and a.shape == (24, 512)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    # decode on the GPU (the stage-1 default): the host only reads the files; a progressive JPEG (decoded on the device too since round 3) and a PNG in the corpus — the PNG takes
    # the PIL detour inside the same call — same pixels either way, so the same embeddings in the same order
    Image.open(paths[3]).save(tmp_path / "prog.jpg", quality=92, progressive=True)
    Image.open(paths[4]).save(tmp_path / "plain.png")
    more = paths + [str(tmp_path / "prog.jpg"), str(tmp_path / "plain.png")]
    d, vd = R.compute_corpus_features(model, None, more, batch=16, gpu_decode=True)
    e, ve = R.compute_corpus_features(model, host_pre, more, batch=5, decode_workers=2)
    assert vd == ve == [p for p in more if "bad" not in p] and d.shape == (26, 512)
    assert np.array_equal(d, e) and np.array_equal(d[:24], a)
    # a pixel budget far below the chunk (a corpus of multi-megapixel files must cost launches, not an OOM abort):
    # the chunk is decoded in many consecutive pieces, every file still lands in its own row with the same bits
    import torch
    n = len(more)
    feats = torch.zeros((n, 512), dtype=torch.float32, device=gpu); okf = torch.zeros((n, 1), dtype=torch.float32, device=gpu)
    st = R._embed_files_gpu_decode(model, more, feats, okf, 16, max_pixels=700 * 500)
    assert st["decode_pieces"] >= 8 and st["failed"] == 1 and st["gpu_decoded"] + st["host_decoded"] == 26
    keep = okf[:, 0].bool().cpu().numpy()
    assert [p for p, k in zip(more, keep) if k] == vd and np.array_equal(feats.cpu().numpy()[keep], d)


def test_style_vectors_of_files_on_the_gpu_equal_the_host_route(gpu, tmp_path):
    """StemStyle.features_from_files (native reads -> GPU JPEG decode -> OpenCV-linear resize kernel -> stem) against
    features_from_path with the same restated resize on the host (PIL decode -> numpy tables): same bits, per file; files the
    device decoder declines (PNG, progressive) and unreadable ones behave like the host route; the re-rank built on either
    route is the same list (compute_resnet_features / resnet_second_stage_rerank, retrieval/…:180-203,454-497)"""
    import numpy as np
    from PIL import Image
    from domain_rag_amd import retrieval as R
    rng = np.random.default_rng(9)
    paths = []
    for i, (h, w) in enumerate([(480, 640), (640, 480), (256, 256), (225, 300), (97, 61), (333, 777), (1000, 400), (64, 64)] * 2):
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
        a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16) + rng.integers(-25, 25, (h, w, 3))
        p = tmp_path / f"{i:03d}.jpg"
        Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(p, quality=int(rng.integers(40, 97)), subsampling=int(rng.integers(0, 3)))
        paths.append(str(p))
    Image.open(paths[0]).save(tmp_path / "plain.png")
    Image.open(paths[1]).save(tmp_path / "prog.jpg", quality=90, progressive=True)
    paths += [str(tmp_path / "plain.png"), str(tmp_path / "prog.jpg"), str(tmp_path / "missing.jpg")]
    stem = R.StemStyle(None, gpu, seed=3)
    got = stem.features_from_files(paths)
    want = [stem.features_from_path(p, restated_resize=True) for p in paths]
    assert got[-1] is None and want[-1] is None
    for g, w_, p in zip(got[:-1], want[:-1], paths):
        assert g is not None and g.shape == (128,) and np.array_equal(g, w_), p
    # the re-rank: device-batched candidates vs one file at a time
    first = [{"similarity": 1.0 - 0.01 * i, "image_path": p, "source_dataset": "coco", "index": i} for i, p in enumerate(paths[1:12])]
    stem.gpu_files = True
    a = R.resnet_second_stage_rerank(paths[0], first, stem, style_cache={})
    stem.gpu_files = False
    stem_host = R.StemStyle(None, gpu, seed=3)
    b = []
    qf = stem_host.features_from_path(paths[0], restated_resize=True)
    for r in first:
        f = stem_host.features_from_path(r["image_path"], restated_resize=True)
        b.append((float(np.linalg.norm(qf - f)), r["image_path"]))
    b.sort(key=lambda t: t[0])
    assert [r["image_path"] for r in a] == [p for _, p in b]
    assert [r["similarity"] for r in a] == [float(1.0 / (1.0 + d)) for d, _ in b]
