"""LaMa stage on the HIP path vs torch's own float32 conv / fft (kernels) and vs the CPU restatement oracle/lama.py (network).
float32 like the reference; stated tolerances: kernels 2e-5 of the output scale (accumulation order only), generator 1e-3
of full scale, inpainted uint8 frame bit-exact outside the mask and within 1 level inside it (the wrapper truncates)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, mode, transposed
    (1, 24, 40, 4, 64, 7, 1, 3, "reflect", False),
    (2, 17, 23, 8, 20, 3, 1, 1, "reflect", False),
    (1, 32, 48, 64, 128, 3, 2, 1, "reflect", False),
    (1, 9, 11, 96, 48, 1, 1, 0, "zero", False),
    (2, 13, 10, 12, 70, 3, 1, 1, "zero", False),
    (1, 8, 12, 32, 16, 3, 2, 1, "zero", True),
    (1, 5, 7, 64, 3, 7, 1, 3, "reflect", False),
    (1, 31, 29, 16, 16, 5, 2, 2, "zero", False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_f32_matches_torch(gpu, case):
    from domain_rag_amd import ops
    B, H, W, Cin, Cout, k, stride, pad, mode, tr = case
    g = torch.Generator().manual_seed(CONV_CASES.index(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    if tr:
        w = torch.randn(Cin, Cout, k, k, generator=g) * 0.2
        ref = F.conv_transpose2d(x, w, stride=stride, padding=pad, output_padding=stride - 1)
        wk = w.permute(1, 2, 3, 0).contiguous()
    else:
        w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
        xp = F.pad(x, (pad,) * 4, mode="reflect") if mode == "reflect" else F.pad(x, (pad,) * 4)
        ref = F.conv2d(xp, w, stride=stride)
        wk = w.permute(0, 2, 3, 1).contiguous()
    Ho, Wo = ref.shape[2:]
    scale, shift = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    addend, resid = torch.randn(B, Cout, Ho, Wo, generator=g), torch.randn(B, Cout, Ho, Wo, generator=g)
    want = F.relu((ref + addend) * scale[None, :, None, None] + shift[None, :, None, None]) + resid
    # channel-slice views: x lives in a wider pixel (ldx = Cin + 8, offset 4), y in a wider one too
    ldx, ldy = Cin + 8, Cout + 5
    xb = torch.zeros(B, H, W, ldx); xb[..., 4:4 + Cin] = _nhwc(x)
    yb = torch.full((B, Ho, Wo, ldy), -7.0)
    xd, yd = xb.to(gpu), yb.to(gpu)
    ops.conv2d_f32(xd.view(-1)[4:], wk.to(gpu), yd.view(-1)[2:], B=B, Hi=H, Wi=W, Ho=Ho, Wo=Wo, Cin=Cin, ldx=ldx, ldy=ldy,
                   stride=stride, pad=pad, pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO, transposed=tr,
                   act=ops.CONV_ACT_RELU, scale=scale.to(gpu), shift=shift.to(gpu), addend=_nhwc(addend).to(gpu), ld_add=Cout,
                   resid=_nhwc(resid).to(gpu), ld_res=Cout)
    got = yd.cpu()
    assert torch.all(got[..., :2] == -7.0) and torch.all(got[..., 2 + Cout:] == -7.0)       # nothing outside the slice
    err = (got[..., 2:2 + Cout] - _nhwc(want)).abs().max().item() / want.abs().max().item()
    assert err < 2e-5, err
    # plain form: no epilogue operands, sigmoid
    y2 = torch.empty(B, Ho, Wo, Cout, device=gpu)
    ops.conv2d_f32(xd.view(-1)[4:], wk.to(gpu), y2, B=B, Hi=H, Wi=W, Ho=Ho, Wo=Wo, Cin=Cin, ldx=ldx, ldy=Cout, stride=stride, pad=pad,
                   pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO, transposed=tr, act=ops.CONV_ACT_SIGMOID)
    assert (y2.cpu() - _nhwc(torch.sigmoid(ref))).abs().max().item() < 2e-5


def test_conv2d_f32_rejects_bad_geometry(gpu):
    from domain_rag_amd import ops
    x = torch.zeros(1, 4, 4, 6, device=gpu); w = torch.zeros(8, 3, 3, 6, device=gpu); y = torch.zeros(1, 4, 4, 8, device=gpu)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        ops.conv2d_f32(x, w, y, B=1, Hi=4, Wi=4, Ho=4, Wo=4, Cin=6, ldx=6, ldy=8, pad=1)
    x = torch.zeros(1, 4, 4, 8, device=gpu); w = torch.zeros(8, 3, 3, 8, device=gpu)
    with pytest.raises(RuntimeError, match="output size"):
        ops.conv2d_f32(x, w, y, B=1, Hi=4, Wi=4, Ho=5, Wo=4, Cin=8, ldx=8, ldy=8, pad=1)
    with pytest.raises(RuntimeError, match="reflect"):
        ops.conv2d_f32(x, torch.zeros(8, 9, 9, 8, device=gpu), y, B=1, Hi=4, Wi=4, Ho=4, Wo=4, Cin=8, ldx=8, ldy=8, pad=4, pad_mode=ops.PAD_REFLECT)


@pytest.mark.parametrize("B,H,W,C", [(1, 8, 8, 64), (2, 7, 9, 24), (1, 63, 47, 20), (1, 12, 10, 192), (1, 1, 2, 4), (1, 3, 1, 4)])
def test_rfft2_irfft2_match_torch(gpu, B, H, W, C):
    from domain_rag_amd import lama, ops
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, H, W, C, generator=g)
    Wf = W // 2 + 1
    tw_w, tw_h = lama._twiddles(W, gpu), lama._twiddles(H, gpu)
    xd = x.to(gpu)
    tmp, f = torch.empty(B, H, Wf, 2 * C, device=gpu), torch.empty(B, H, Wf, 2 * C, device=gpu)
    ops.rfft2_f32(xd, tmp, f, B, H, W, C, C, tw_w, tw_h)
    ref = torch.fft.rfftn(x.permute(0, 3, 1, 2).double(), dim=(-2, -1), norm="ortho")               # [B,C,H,Wf]
    ref_il = torch.stack((ref.real, ref.imag), dim=-1).permute(0, 2, 3, 1, 4).reshape(B, H, Wf, 2 * C)     # channel 2c | 2c+1
    scale = ref_il.abs().max().item()
    assert (f.cpu().double() - ref_il).abs().max().item() < 2e-5 * scale
    # inverse on an ARBITRARY spectrum (what the 1x1 conv produces: DC / Nyquist bins carry imaginary parts that c2r ignores)
    spec = torch.randn(B, H, Wf, 2 * C, generator=g)
    add = torch.randn(B, H, W, C, generator=g)
    y = torch.empty(B, H, W, C, device=gpu)
    ops.irfft2_f32(spec.to(gpu), tmp, y, add.to(gpu), B, H, W, C, C, C, tw_w, tw_h)
    sc = spec.view(B, H, Wf, C, 2).permute(0, 3, 1, 2, 4).double()
    inv = torch.fft.irfftn(torch.complex(sc[..., 0].contiguous(), sc[..., 1].contiguous()), s=(H, W), dim=(-2, -1), norm="ortho")
    want = inv.permute(0, 2, 3, 1) + add.double()
    assert (y.cpu().double() - want).abs().max().item() < 2e-5 * want.abs().max().item()


def _ocfg(cfg):
    from oracle import lama as olama
    return olama.LamaConfig(ngf=cfg.ngf, n_down=cfg.n_down, n_blocks=cfg.n_blocks, ratio_g=cfg.ratio_g, bn_eps=cfg.bn_eps)


def _image_and_mask(H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    img = (img.astype(np.int32) + rng.integers(-20, 20, img.shape)).clip(0, 255).astype(np.uint8)
    mask = np.zeros((H, W), np.uint8)
    mask[H // 4: H // 4 + H // 3, W // 3: W // 3 + W // 4] = 255
    mask[H - 3:, W - 5:] = 1                                   # any non-zero value fills; touches the padded corner
    return img, mask


@pytest.mark.parametrize("H,W", [(64, 64), (61, 83), (40, 24)])
def test_tiny_generator_and_frame_vs_oracle(gpu, H, W):
    from domain_rag_amd import lama
    from oracle import lama as olama
    cfg = lama.LamaConfig(ngf=16, n_blocks=3)
    p = lama.init_params(cfg, seed=5)
    net = lama.LamaHIP(cfg, p, gpu)
    img, mask = _image_and_mask(H, W, 1)
    out = net(torch.from_numpy(img).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    ti, tm = olama.prepare_img_and_mask(img, mask)
    pred_ref = olama.generator(p, _ocfg(cfg), torch.cat([ti * (1 - tm), tm], 1))
    Hp, Wp = ti.shape[2:]
    assert out.shape == (Hp, Wp, 3) and (Hp % 8, Wp % 8) == (0, 0) and Hp - H < 8 and Wp - W < 8
    pred = net._bufs["pred"][..., :3].cpu()
    err = (pred - pred_ref[0].permute(1, 2, 0)).abs().max().item()
    assert err < 1e-3, err                                                # sigmoid outputs in (0,1): absolute = of full scale
    assert 0.02 < pred_ref.std().item()                                   # the synthetic network is not saturated flat
    ref = olama.inpaint(p, _ocfg(cfg), img, mask)
    m = tm[0, 0].numpy() > 0
    assert np.array_equal(out[~m], ref[~m])                               # kept pixels: the /255 *255 truncation chain, bit for bit
    d = np.abs(out[m].astype(np.int32) - ref[m].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02
    # float32 (u/255)*255 truncates back to u for all 256 levels: the kept region is the (symmetrically padded) input
    padded = np.pad(img, ((0, Hp - H), (0, Wp - W), (0, 0)), mode="symmetric")
    assert np.array_equal(padded[~m], out[~m])


def test_big_lama_architecture_vs_oracle(gpu):
    """the real configuration (ngf 64, 18 blocks, 128 | 384 channels) on a 136x200 frame"""
    from domain_rag_amd import lama
    from oracle import lama as olama
    cfg = lama.LamaConfig()
    p = lama.init_params(cfg, seed=2)
    net = lama.LamaHIP(cfg, p, gpu)
    img, mask = _image_and_mask(133, 200, 3)
    out = net(torch.from_numpy(img).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    ref = olama.inpaint(p, _ocfg(cfg), img, mask)
    ti, tm = olama.prepare_img_and_mask(img, mask)
    m = tm[0, 0].numpy() > 0
    assert out.shape == ref.shape == (136, 200, 3)
    assert np.array_equal(out[~m], ref[~m])
    d = np.abs(out[m].astype(np.int32) - ref[m].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02
    assert ref[m].std() > 5                                               # a real picture inside the hole, not a constant
    # size history: another frame size and back gives the same bytes
    img2, mask2 = _image_and_mask(64, 72, 4)
    net(torch.from_numpy(img2).to(gpu), torch.from_numpy(mask2).to(gpu))
    again = net(torch.from_numpy(img).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    assert np.array_equal(out, again)


def test_simple_lama_dropin(gpu, monkeypatch):
    from PIL import Image
    from domain_rag_amd import lama
    monkeypatch.setenv("DRAG_SYNTHETIC_WEIGHTS", "1"); monkeypatch.setenv("DRAG_TINY", "1")
    sl = lama.SimpleLama()
    img, mask = _image_and_mask(50, 70, 9)
    res = sl(Image.fromarray(img), Image.fromarray(mask))
    assert isinstance(res, Image.Image) and res.size == (72, 56) and res.mode == "RGB"
    with pytest.raises(RuntimeError, match="expected input.*channels"):      # the branch lama_inpaint.py:176-178 keys on
        sl(Image.fromarray(img[..., 0]), Image.fromarray(mask))
    monkeypatch.delenv("DRAG_SYNTHETIC_WEIGHTS")
    monkeypatch.setenv("LAMA_MODEL", "/nonexistent/big-lama.pt")
    with pytest.raises(FileNotFoundError):
        lama.SimpleLama()


def test_stage0_cli_on_mini_dataset(gpu, tmp_path):
    """python -m domain_rag_amd.cli.stage0_lama from ./lama_inpaint like the reference's script: output tree, sizes, logs;
    kept pixels of a PNG output are the input's, the hole is repainted; two torchrun-style ranks write the same files."""
    import json
    import os
    import subprocess
    import sys
    from PIL import Image
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(3)
    ds = tmp_path / "datasets" / "NEU-DET"
    (ds / "train").mkdir(parents=True); (ds / "annotations").mkdir()
    (tmp_path / "lama_inpaint").mkdir()
    images, anns = [], []
    for i, (name, h, w) in enumerate([("a.png", 40, 56), ("b.jpg", 37, 50), ("c.png", 64, 64)]):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(arr).save(ds / "train" / name)
        images.append({"id": i + 1, "file_name": name, "width": w, "height": h})
        anns.append({"id": i + 1, "image_id": i + 1, "bbox": [8 + i, 6, 20.5, 14], "category_id": 1})
    anns.append({"id": 9, "image_id": 1, "bbox": [40, 30, 100, 100], "category_id": 2})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "crazing"}, {"id": 2, "name": "patches"}]},
              open(ds / "annotations" / "1_shot.json", "w"))

    def run(extra_env=None):
        env = dict(os.environ, PYTHONPATH=ROOT, DRAG_TIMESTAMP="20260101_000000", **(extra_env or {}))
        r = subprocess.run([sys.executable, "-m", "domain_rag_amd.cli.stage0_lama", "--datasets", "NEU-DET", "missing_ds", "--shots", "1",
                            "--synthetic-weights", "--tiny"], cwd=tmp_path / "lama_inpaint", env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout + r.stderr

    log = run()
    out = tmp_path / "lamainpaint" / "NEU_DET" / "1_shot"
    assert "完成 3 个图像, 失败 0 个" in log and "多bbox图像 a.png: 2 个bbox" in log and "数据集目录不存在: ../datasets/missing_ds" in log
    assert (tmp_path / "lamainpaint" / "logs" / "lama_inpaint_20260101_000000.log").exists()
    assert Image.open(out / "a.png").size == (56, 40) and Image.open(out / "b.jpg").size == (56, 40) and Image.open(out / "c.png").size == (64, 64)
    from domain_rag_amd import hostlogic as H
    src, res = np.asarray(Image.open(ds / "train" / "a.png")), np.asarray(Image.open(out / "a.png"))
    m = H.inpaint_mask_array(56, 40, [a["bbox"] for a in anns if a["image_id"] == 1]) > 0
    assert np.array_equal(src[~m], res[~m]) and (src[m] != res[m]).mean() > 0.9
    first = {n: np.asarray(Image.open(out / n)).copy() for n in ("a.png", "b.jpg", "c.png")}
    for n in first:
        os.remove(out / n)
    run({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert sorted(os.listdir(out)) == ["a.png", "b.jpg"]                         # contiguous split of 3 images: [2, 1]
    run({"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"})                        # LOCAL_RANK beyond the device count wraps
    for n, a in first.items():
        assert np.array_equal(np.asarray(Image.open(out / n)), a)


def test_reference_style_import_through_compat(gpu, monkeypatch):
    """``from simple_lama_inpainting import SimpleLama`` as lama_inpaint.py:5 writes it, after compat.install()"""
    import sys
    from PIL import Image
    from domain_rag_amd import compat
    monkeypatch.setenv("DRAG_SYNTHETIC_WEIGHTS", "1"); monkeypatch.setenv("DRAG_TINY", "1")
    saved = {k: sys.modules.get(k) for k in ("simple_lama_inpainting",)}
    try:
        compat.install(("simple_lama_inpainting",))
        from simple_lama_inpainting import SimpleLama
        img, mask = _image_and_mask(32, 40, 2)
        assert SimpleLama()(Image.fromarray(img), Image.fromarray(mask)).size == (40, 32)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_conv_tile_variants_give_the_same_bits(gpu):
    """the conv picks its tile by launch size (64x64 with 64-channel steps for a handful of workgroups, 64x64 with 16-channel
    steps, 128x128 for big launches with >= 128 output channels); all walk the taps and channels in the same order, so a
    pixel's value must not depend on how many images share the launch"""
    from domain_rag_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 16, 128, generator=g).to(gpu)
    geo = dict(Hi=16, Wi=16, Ho=16, Wo=16, Cin=128, ldx=128, pad=1, pad_mode=ops.PAD_REFLECT)
    xb = x.expand(300, 16, 16, 128).contiguous()
    for cout in (64, 160):          # 64: 4 workgroups (64-ch steps) vs 1200 (16-ch steps); 160: 12 (64-ch steps) vs 600x2 tiles of 128x128
        w = (torch.randn(cout, 3, 3, 128, generator=g) * 0.1).to(gpu)
        sc, sh = torch.randn(cout, generator=g).to(gpu), torch.randn(cout, generator=g).to(gpu)
        y1 = torch.empty(1, 16, 16, cout, device=gpu)
        ops.conv2d_f32(x, w, y1, B=1, ldy=cout, scale=sc, shift=sh, act=ops.CONV_ACT_RELU, **geo)
        yb = torch.empty(300, 16, 16, cout, device=gpu)
        ops.conv2d_f32(xb, w, yb, B=300, ldy=cout, scale=sc, shift=sh, act=ops.CONV_ACT_RELU, **geo)
        assert torch.equal(yb[0], y1[0]) and torch.equal(yb[299], y1[0]) and torch.equal(yb[150], y1[0])
        ref = F.relu(F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1,) * 4, mode="reflect"), w.permute(0, 3, 1, 2)) * sc[None, :, None, None] + sh[None, :, None, None])
        assert (y1.permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_full_config_frame_properties(gpu):
    """size-independent properties at a real frame size (640x480, big-lama configuration), where the CPU oracle is slow: an empty
    mask returns the padded input untouched, kept pixels are the input's whatever the hole, the hole is repainted, and the
    result does not depend on what the hole contained (the generator only ever sees img * (1 - mask))"""
    from domain_rag_amd import lama
    cfg = lama.LamaConfig()
    net = lama.LamaHIP(cfg, lama.init_params(cfg, seed=4), gpu)
    img, mask = _image_and_mask(477, 635, 6)
    Hp, Wp = 480, 640
    padded = np.pad(img, ((0, Hp - 477), (0, Wp - 635), (0, 0)), mode="symmetric")
    dev = lambda a: torch.from_numpy(a).to(gpu)
    empty = net(dev(img), dev(np.zeros_like(mask))).cpu().numpy()
    assert empty.shape == (Hp, Wp, 3) and np.array_equal(empty, padded)
    out = net(dev(img), dev(mask)).cpu().numpy()
    m = np.pad(mask, ((0, Hp - 477), (0, Wp - 635)), mode="symmetric") > 0
    assert np.array_equal(out[~m], padded[~m]) and (out[m] != padded[m]).mean() > 0.9
    scribbled = img.copy()
    scribbled[mask > 0] = 255 - scribbled[mask > 0]
    out2 = net(dev(scribbled), dev(mask)).cpu().numpy()
    assert np.array_equal(out2[m], out[m])
    full = net(dev(img), dev(np.full_like(mask, 255))).cpu().numpy()          # everything is hole: pure generator output
    assert full.shape == (Hp, Wp, 3) and full.std() > 1
