"""LaMa stage, CPU side: the host logic against goldens captured from the imported reference script
(tests/golden/make_lama_goldens.py -> lama_stage.json) and self-checks of the CPU restatement of the network."""
import hashlib
import importlib.util
import json
import logging
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "lama_stage.json")))


def _gen():
    spec = importlib.util.spec_from_file_location("make_lama_goldens", os.path.join(HERE, "golden", "make_lama_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_inpaint_masks_match_reference_goldens():
    from domain_rag_amd import hostlogic as H
    gen = _gen()
    for c in GOLD["masks"]:
        m = H.inpaint_mask_array(c["w"], c["h"], c["boxes"])
        assert m.shape == (c["h"], c["w"]) and m.dtype == np.uint8
        assert gen.rle_rows(m) == c["runs"], c["boxes"]
        assert sorted(int(v) for v in np.unique(m)) == c["values"]
    for c in GOLD["random_masks"]:
        m = H.inpaint_mask_array(c["w"], c["h"], c["boxes"])
        assert hashlib.sha1(m.tobytes()).hexdigest() == c["sha1"], c


def test_process_dataset_flow_matches_reference(tmp_path, monkeypatch):
    """same crafted dataset, same recording stand-in for the model: what the model is handed, what is written where, the counters"""
    from PIL import Image
    from domain_rag_amd.cli import stage0_lama as s0
    gen = _gen()
    gen.write_dataset(str(tmp_path))
    record = []

    def model(image, mask):
        a, m = np.asarray(image), np.asarray(mask)
        record.append(["call", image.mode, list(image.size), mask.mode, list(mask.size), hashlib.sha1(a.tobytes()).hexdigest(),
                       gen.rle_rows(m), sorted(int(v) for v in np.unique(m))])
        if a[0, 0, 0] == 7:
            raise ValueError("boom")
        return Image.fromarray(255 - a)

    monkeypatch.chdir(tmp_path / "lama_inpaint")
    logger = logging.getLogger("test_lama_stage"); logger.addHandler(logging.NullHandler()); logger.propagate = False
    counts = s0.process_dataset("NEU-DET", "1", logger, model)
    missing = s0.process_dataset("nope", "1", logger, model)
    g = GOLD["dataset"]
    assert list(counts) == g["counts"] and list(missing) == g["missing_dataset_counts"]
    assert record == [c for c in g["calls"] if c[0] == "call"]           # incl. the bicubic resize to the annotated size (sha1 of the pixels)
    outs = {}
    base = tmp_path / "lamainpaint"
    for dp, _, fs in os.walk(base):
        for f in fs:
            if f.endswith(".log"):
                continue
            full = os.path.join(dp, f)
            outs[os.path.relpath(full, base)] = [list(Image.open(full).size), Image.open(full).mode,
                                                 hashlib.sha1(np.asarray(Image.open(full)).tobytes()).hexdigest()]
    assert outs == g["outputs"]                                            # NEU-DET -> NEU_DET, nested file names kept
    # sharded: two ranks together process exactly the same images, contiguous split in annotation order
    record.clear()
    c0 = s0.process_dataset("NEU-DET", "1", logger, model, rank=0, world=2)
    n0 = len(record)
    c1 = s0.process_dataset("NEU-DET", "1", logger, model, rank=1, world=2)
    assert record == [c for c in g["calls"] if c[0] == "call"] and 0 < n0 < len(record)
    assert [c0[0] + c1[0], c0[1] + c1[1]] == g["counts"]


def test_lama_output_dir_and_parser_defaults():
    from domain_rag_amd import hostlogic as H
    from domain_rag_amd.cli import stage0_lama as s0
    assert H.lama_output_dir("NEU-DET", "5") == "../lamainpaint/NEU_DET/5_shot"
    assert H.lama_output_dir("my set", 1) == "../lamainpaint/my_set/1_shot"
    a = s0.build_parser().parse_args([])
    assert a.datasets == ["ArTaxOr", "clipart1k", "DIOR", "FISH", "NEU-DET"] and a.shots == ["1", "2", "3", "5", "10"] and not a.fix_channels


def test_oracle_fourier_unit_is_the_definition():
    """the restated FourierUnit against an explicit float64 DFT-matrix evaluation of rfft2 -> 1x1 conv + BN + ReLU -> irfft2"""
    from oracle import lama as ol
    g = torch.Generator().manual_seed(0)
    B, C, H, W = 1, 3, 6, 5
    x = torch.randn(B, C, H, W, generator=g)
    p = {"fu.conv_layer.weight": torch.randn(2 * C, 2 * C, 1, 1, generator=g) * 0.5, "fu.bn.weight": torch.rand(2 * C, generator=g) + 0.5,
         "fu.bn.bias": torch.randn(2 * C, generator=g) * 0.1, "fu.bn.running_mean": torch.randn(2 * C, generator=g) * 0.1,
         "fu.bn.running_var": torch.rand(2 * C, generator=g) + 0.5}
    got = ol.fourier_unit(p, "fu", x, 1e-5)
    xd = x.double().numpy()[0]
    Wf = W // 2 + 1
    FH = np.exp(-2j * np.pi * np.outer(np.arange(H), np.arange(H)) / H)
    FW = np.exp(-2j * np.pi * np.outer(np.arange(W), np.arange(Wf)) / W)
    spec = np.einsum("kh,chw,wl->ckl", FH, xd, FW) / np.sqrt(H * W)                       # [C,H,Wf]
    chans = np.stack([spec.real, spec.imag], 1).reshape(2 * C, H, Wf)                     # 2c = re, 2c+1 = im
    wgt = p["fu.conv_layer.weight"].double().numpy()[:, :, 0, 0]
    y = np.einsum("oc,chw->ohw", wgt, chans)
    s = (p["fu.bn.weight"].double() / torch.sqrt(p["fu.bn.running_var"].double() + 1e-5)).numpy()
    y = np.maximum((y - p["fu.bn.running_mean"].double().numpy()[:, None, None]) * s[:, None, None] + p["fu.bn.bias"].double().numpy()[:, None, None], 0)
    z = y.reshape(C, 2, H, Wf)
    z = z[:, 0] + 1j * z[:, 1]
    # c2r: Hermitian extension along W; the imaginary parts of the DC (and, for even W, Nyquist) columns do not contribute
    full = np.zeros((C, H, W), complex)
    zh = np.einsum("hk,ckl->chl", np.conj(FH), z)                                         # inverse along H first (not Hermitian-symmetrised)
    out = np.zeros((C, H, W))
    for l in range(Wf):
        wl = 1.0 if l == 0 or (W % 2 == 0 and l == W // 2) else 2.0
        ph = np.exp(2j * np.pi * l * np.arange(W) / W)
        v = zh[:, :, l].copy()
        if wl == 1.0:
            v = v.real + 0j
        out += wl * (v[:, :, None] * ph[None, None, :]).real
    out /= np.sqrt(H * W)
    assert np.abs(got.double().numpy()[0] - out).max() < 1e-5
    del full


def test_oracle_generator_shapes_and_frame_rules():
    from domain_rag_amd import lama
    from oracle import lama as ol
    cfg = lama.LamaConfig(ngf=8, n_blocks=1)
    p = lama.init_params(cfg, 0)
    oc = ol.LamaConfig(ngf=8, n_blocks=1)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (21, 30, 3), dtype=np.uint8)
    mask = np.zeros((21, 30), np.uint8); mask[5:12, 8:20] = 200; mask[20, 29] = 1
    out = ol.inpaint(p, oc, img, mask)
    assert out.shape == (24, 32, 3) and out.dtype == np.uint8                       # padded frame is returned as it is
    ti, tm = ol.prepare_img_and_mask(img, mask)
    assert ti.shape == (1, 3, 24, 32) and set(np.unique(tm.numpy())) == {0.0, 1.0}
    assert np.array_equal(ti[0, :, 21:, :].numpy(), ti[0, :, 18:21, :].flip(1).numpy())      # symmetric: edge row repeated, then inwards
    assert tm[0, 0, 21, 30] == 1 and tm[0, 0, 20, 30] == 1 and tm[0, 0, 21, 31] == 0                            # mask padded the same way, binarised with > 0
    keep = tm[0, 0].numpy() == 0
    padded = np.pad(img, ((0, 3), (0, 2), (0, 0)), mode="symmetric")
    assert np.array_equal(padded[keep], out[keep])            # (u/255)*255 in float32 truncates back to u for every level: kept pixels are the input
    # parameter bookkeeping of the real configuration: the published big-lama has 51 M generator parameters
    big = lama.init_params(lama.LamaConfig(), 0)
    n = sum(v.numel() for k, v in big.items() if "running" not in k)
    assert 50.5e6 < n < 51.5e6
    sd = {"model.generator." + k: v for k, v in big.items()}
    sd["model.generator.model.1.bn_l.num_batches_tracked"] = torch.tensor(3)
    norm = lama.normalize_state_dict(sd)
    assert set(norm) == set(big)
    c = lama.config_from_state_dict(norm)
    assert (c.ngf, c.n_down, c.n_blocks, c.c_local, c.c_global) == (64, 3, 18, 128, 384)


def test_lama_needs_the_gpu():
    from domain_rag_amd import lama
    with pytest.raises(RuntimeError, match="no CPU path"):
        lama.LamaHIP(lama.LamaConfig(ngf=8, n_blocks=1), lama.init_params(lama.LamaConfig(ngf=8, n_blocks=1), 0), "cpu")


def test_weight_files_load_through_both_containers(tmp_path):
    """big-lama.pt is a TorchScript export whose parameters sit under some wrapper prefix; a training checkpoint keeps them
    under ``generator.``.  Both must come back as the generator's own ``model.<i>.…`` names with the same values."""
    import torch.nn as nn
    from domain_rag_amd import lama
    cfg = lama.LamaConfig(ngf=8, n_blocks=1)
    p = lama.init_params(cfg, 0)
    root = nn.Module()
    for k, v in p.items():
        parts = ("model.generator." + k).split(".")
        m = root
        for a in parts[:-1]:
            if not hasattr(m, a):
                m.add_module(a, nn.Module())
            m = getattr(m, a)
        if "running" in parts[-1]:
            m.register_buffer(parts[-1], v.clone())
        else:
            m.register_parameter(parts[-1], nn.Parameter(v.clone()))
    torch.jit.save(torch.jit.script(root), str(tmp_path / "big-lama.pt"))
    sd = lama.load_state_dict(str(tmp_path / "big-lama.pt"))
    assert set(sd) == set(p) and all(torch.equal(sd[k], p[k]) for k in p)
    torch.save({"state_dict": {"generator." + k: v for k, v in p.items()}, "epoch": 3}, str(tmp_path / "best.ckpt"))
    sd2 = lama.load_state_dict(str(tmp_path / "best.ckpt"))
    assert set(sd2) == set(p) and all(torch.equal(sd2[k], p[k]) for k in p)
    c = lama.config_from_state_dict(sd)
    assert (c.ngf, c.n_down, c.n_blocks, c.ratio_g) == (8, 3, 1, 0.75)
    torch.save({"state_dict": {"discriminator.x": torch.zeros(1)}}, str(tmp_path / "other.ckpt"))
    with pytest.raises(RuntimeError, match="no FFCResNetGenerator parameters"):
        lama.load_state_dict(str(tmp_path / "other.ckpt"))
