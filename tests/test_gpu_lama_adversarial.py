"""LaMa kernels outside the comfortable input distribution (DESIGN "Known gaps after round 4": the stage's kernels had
only seen N(0, 1) operands and one synthetic picture).  Same pattern as the hot-key attention and large-mean GroupNorm
tests: inputs on which an accumulation-order or boundary mistake shows, each checked against float64 / the oracle with
torch's own float32 result as the yardstick (a float32 kernel may be as wrong as float32 arithmetic is, not more).

* prepare / blend: every uint8 level, every mask value class, predictions ON the truncation boundaries (k / 255 and its
  float32 neighbours), out of range, signed zero, huge — bit-exact against the reference expression
  (oracle/lama.py prepare_img_and_mask / inpaint: `m * pred + (1 - m) * img`, `clip(x * 255, 0, 255).astype(uint8)`),
  on frames whose symmetric padding is WIDER than the picture (several reflections).
* conv2d_f32: operands with a large common mean and weights that cancel it (the sum of products is 1e-4 of the sum of
  magnitudes), a lone hot pixel, folded-BatchNorm scales spanning 1e-6 .. 1e6.
* rfft2 / irfft2: a DC level 1e4 above the signal, a delta, a pure Nyquist checkerboard, a spectrum with one hot bin.
* the generator on all-masked, unmasked, black and white pictures and on the smallest frame the reference accepts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------ prepare / blend: integer and boundary behaviour
def _boundary_predictions(n, seed):
    """float32 values around every truncation boundary of `clip(v * 255, 0, 255).astype(uint8)`"""
    k = np.arange(256, dtype=np.float32)
    on = k / np.float32(255)
    vals = np.concatenate([
        on, np.nextafter(on, np.float32(-1)), np.nextafter(on, np.float32(2)),
        (k + np.float32(0.5)) / np.float32(255),
        np.array([0.0, -0.0, 1.0, np.nextafter(np.float32(1), np.float32(2)), -1e-30, 1e-45, -1e-45, 1.5, -3.0, 1e30, -1e30,
                  256.0 / 255.0, 0.99999994, 3.4e38, -3.4e38], dtype=np.float32),
    ]).astype(np.float32)
    rng = np.random.default_rng(seed)
    return vals[rng.integers(0, len(vals), n)]


@pytest.mark.parametrize("H,W", [(5, 3), (1, 1), (16, 16), (9, 23), (2, 7), (8, 1)])
def test_prepare_and_blend_bit_exact_on_boundary_values(gpu, H, W):
    from domain_rag_amd import ops
    from oracle import lama as olama
    rng = np.random.default_rng(H * 31 + W)
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    img.reshape(-1)[:min(256, img.size)] = np.arange(min(256, img.size), dtype=np.uint8)      # every level where there is room
    mask = rng.choice(np.array([0, 0, 0, 1, 2, 127, 128, 254, 255], np.uint8), (H, W))
    Hp, Wp = -(-H // 8) * 8, -(-W // 8) * 8
    ti, tm = olama.prepare_img_and_mask(img, mask)                         # [1,3,Hp,Wp], [1,1,Hp,Wp]
    assert ti.shape[2:] == (Hp, Wp)
    x_ref = _nhwc(torch.cat([ti * (1 - tm), tm], 1))[0]
    imd, mkd = torch.from_numpy(img).to(gpu), torch.from_numpy(mask).to(gpu)
    x0 = torch.full((Hp, Wp, 4), -5.0, device=gpu)
    ops.lama_prepare(imd, mkd, x0, H, W, Hp, Wp)
    assert torch.equal(x0.cpu().view(torch.int32), x_ref.contiguous().view(torch.int32))       # bits, signed zeros included
    for ld in (3, 4, 7):
        pred = torch.from_numpy(_boundary_predictions(Hp * Wp * 3, H + W + ld).reshape(1, 3, Hp, Wp))
        out_ref = tm * pred + (1 - tm) * ti
        with np.errstate(over="ignore"):                                   # 3.4e38 * 255 = inf, clipped to 255 as the reference does
            ref = np.clip(out_ref[0].permute(1, 2, 0).numpy() * 255, 0, 255).astype(np.uint8)
        pb =torch.full((Hp, Wp, ld), 0.25)
        pb[..., :3] = _nhwc(pred)[0]
        out = torch.full((Hp, Wp, 3), 77, dtype=torch.uint8, device=gpu)
        ops.lama_blend(pb.to(gpu), ld, imd, mkd, out, H, W, Hp, Wp)
        got = out.cpu().numpy()
        assert np.array_equal(got, ref), (ld, np.argwhere(got != ref)[:5])
        m = tm[0, 0].numpy() > 0
        padded = np.pad(img, ((0, Hp - H), (0, Wp - W), (0, 0)), mode="symmetric")
        assert np.array_equal(got[~m], padded[~m])                         # kept pixels are the input's bytes


# ------------------------------------------------------------------ conv2d_f32: cancellation, hot pixels, extreme BN scales
def _conv_call(ops, gpu, x, w, *, stride, pad, mode, **epi):
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, device=gpu)
    ops.conv2d_f32(_nhwc(x).to(gpu), w.permute(0, 2, 3, 1).contiguous().to(gpu), y, B=B, Hi=H, Wi=W, Ho=Ho, Wo=Wo, Cin=Cin,
                   ldx=Cin, ldy=Cout, stride=stride, pad=pad, pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO, **epi)
    return y.cpu().permute(0, 3, 1, 2)


def _pad(x, pad, mode):
    return F.pad(x, (pad,) * 4, mode="reflect") if mode == "reflect" else F.pad(x, (pad,) * 4)


ADV_CONV = [
    # Cin, Cout, k, stride, pad, mode, H, W
    (64, 64, 3, 1, 1, "reflect", 20, 28),
    (4, 64, 7, 1, 3, "reflect", 18, 26),
    (128, 96, 1, 1, 0, "zero", 12, 20),
    (64, 128, 3, 2, 1, "zero", 21, 33),
    (64, 3, 7, 1, 3, "reflect", 16, 24),
]


@pytest.mark.parametrize("case", ADV_CONV)
def test_conv2d_f32_cancelling_sums_stay_at_float32_accuracy(gpu, case):
    """x = 1000 + N(0, 1), every filter's weights sum to zero: the result is 1e-4 of the sum of magnitudes.  A float32
    accumulation errs by ~eps * sum |x||w| whatever its order; the kernel may err as much as torch's float32 conv does
    (4x slack for the different order), not more."""
    from domain_rag_amd import ops
    Cin, Cout, k, stride, pad, mode, H, W = case
    g = torch.Generator().manual_seed(ADV_CONV.index(case) + 40)
    x = 1000.0 + torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    w = w - w.mean(dim=(1, 2, 3), keepdim=True)
    xp = _pad(x, pad, mode)
    ref64 = F.conv2d(xp.double(), w.double(), stride=stride)
    cond = F.conv2d(xp.abs().double(), w.abs().double(), stride=stride)
    e_torch = ((F.conv2d(xp, w, stride=stride).double() - ref64).abs() / cond).max().item()
    got = _conv_call(ops, gpu, x, w, stride=stride, pad=pad, mode=mode)
    e_hip = ((got.double() - ref64).abs() / cond).max().item()
    assert e_hip <= max(4 * e_torch, 4e-7), (e_hip, e_torch)              # 4e-7 = 6 ulp of the magnitude sum
    if mode == "reflect":                                                 # (a zero border breaks the cancellation by design)
        assert ref64.abs().max().item() < 1e-2 * cond.max().item()        # the case really cancels


def test_conv2d_f32_hot_pixel_and_extreme_batchnorm_scales(gpu):
    """one pixel 1e6 among N(0, 1): outputs away from it must not be polluted (masked / out-of-window lanes contribute an
    exact 0, not 0 * 1e6 after a rounding); folded-BN scale / shift from 1e-6 to 1e6 per channel; ReLU on exact zeros"""
    from domain_rag_amd import ops
    g = torch.Generator().manual_seed(77)
    Cin, Cout, H, W = 32, 48, 19, 27
    x = torch.randn(1, Cin, H, W, generator=g)
    x[0, 5, 9, 13] = 1e6
    x[0, 6, 1, 1] = -1e6                                                   # reflected into the border (four copies in the padded frame)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    scale = 10.0 ** torch.linspace(-6, 6, Cout) * torch.where(torch.arange(Cout) % 2 == 0, 1.0, -1.0)
    shift = torch.randn(Cout, generator=g) * scale.abs()
    xp = _pad(x, 1, "reflect")
    conv64 = F.conv2d(xp.double(), w.double())
    want = F.relu(conv64 * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    cond = (F.conv2d(xp.abs().double(), w.abs().double()) * scale.abs().double()[None, :, None, None]
            + shift.abs().double()[None, :, None, None])
    got = _conv_call(ops, gpu, x, w, stride=1, pad=1, mode="reflect", act=ops.CONV_ACT_RELU, scale=scale.to(gpu), shift=shift.to(gpu))
    assert torch.isfinite(got).all()
    # a window that holds the 1e6 pixel rounds every later partial sum at ulp(2e5): torch's own float32 conv sits at 1.5e-6 of
    # the magnitude sum there (this kernel: 1.3e-6 on the first run) — the yardstick is torch, not a fixed number of ulps
    t32 = F.relu(F.conv2d(xp, w) * scale[None, :, None, None] + shift[None, :, None, None])
    e_torch = ((t32.double() - want).abs() / cond).max().item()
    e_hip = ((got.double() - want).abs() / cond).max().item()
    assert e_hip <= max(2 * e_torch, 4e-7), (e_hip, e_torch)
    # far from both hot pixels the window holds N(0, 1) values only: error relative to THOSE magnitudes
    far = torch.ones(H, W, dtype=torch.bool)
    far[7:12, 11:16] = False
    far[0:3, 0:3] = False
    local = F.conv2d(xp.abs().double(), w.abs().double())
    assert local[0][:, far].max().item() < 1e3                             # the selection really excludes the hot windows
    conv32 = _conv_call(ops, gpu, x, w, stride=1, pad=1, mode="reflect")
    assert ((conv32.double() - conv64).abs() / local)[0][:, far].max().item() < 4e-7


# ------------------------------------------------------------------ rfft2 / irfft2 on spectra with one dominant term
def _rfft2(ops, lama, gpu, x):
    B, H, W, C = x.shape
    Wf = W // 2 + 1
    tmp, f = torch.empty(B, H, Wf, 2 * C, device=gpu), torch.empty(B, H, Wf, 2 * C, device=gpu)
    ops.rfft2_f32(x.to(gpu), tmp, f, B, H, W, C, C, lama._twiddles(W, gpu), lama._twiddles(H, gpu))
    return f.cpu()


def _interleave(c):                                                        # complex [B,C,H,Wf] -> [B,H,Wf,2C] (re | im per channel)
    B, C, H, Wf = c.shape
    return torch.stack((c.real, c.imag), dim=-1).permute(0, 2, 3, 1, 4).reshape(B, H, Wf, 2 * C)


@pytest.mark.parametrize("H,W,C", [(32, 32, 16), (25, 38, 8), (64, 48, 4)])
@pytest.mark.parametrize("kind", ["dc", "delta", "nyquist", "row"])
def test_rfft2_with_one_dominant_term(gpu, H, W, C, kind):
    """An orthonormal transform in float32 errs by ~eps * ||x||_2 per bin (a DFT by matrix products with a float32 twiddle
    table 1.1-5x an FFT's error, 1e-8..3e-7 ||x||_2 — emulated on the CPU with the same table): bar = 6x torch's own float32
    FFT error or 6e-7 ||x||_2.  The small bins next to a 1e4 DC level are where a
    twiddle table with one wrong entry, or a DC row that is not exactly 1, shows."""
    from domain_rag_amd import lama, ops
    g = torch.Generator().manual_seed(H + W + C)
    x = torch.randn(1, H, W, C, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    if kind == "dc":
        x = x + 1e4
    elif kind == "delta":
        x[0, H // 3, W // 5, :] = 1e6
    elif kind == "nyquist":
        x = x + 1e4 * ((-1.0) ** (yy + xx))[None, :, :, None]
    else:
        x[0, H // 2, :, :] += 1e5                                          # one bright row: a flat line in the spectrum
    xc = x.permute(0, 3, 1, 2)
    ref = _interleave(torch.fft.rfftn(xc.double(), dim=(-2, -1), norm="ortho"))
    t32 = _interleave(torch.fft.rfftn(xc, dim=(-2, -1), norm="ortho")).double()
    got = _rfft2(ops, lama, gpu, x).double()
    norm = xc.double().flatten(2).norm(dim=2).max().item()
    e_hip, e_t = (got - ref).abs().max().item(), (t32 - ref).abs().max().item()
    assert e_hip <= max(6 * e_t, 6e-7 * norm), (e_hip, e_t, norm)


@pytest.mark.parametrize("H,W,C", [(32, 32, 16), (25, 38, 8)])
def test_irfft2_with_a_hot_bin_and_imaginary_dc(gpu, H, W, C):
    """c2r ignores the imaginary parts of the self-conjugate bins (DC; Nyquist for even sizes) — here they carry 1e6 — and one
    interior bin 1e5 above the rest must come back as a clean plane wave"""
    from domain_rag_amd import lama, ops
    g = torch.Generator().manual_seed(H * W + C)
    Wf = W // 2 + 1
    spec = torch.randn(1, H, Wf, C, 2, generator=g)
    spec[0, 0, 0, :, 1] = 1e6                                              # Im(DC)
    if W % 2 == 0:
        spec[0, 0, Wf - 1, :, 1] = -1e6                                    # Im(Nyquist column, row 0)
    spec[0, 3, 2, :, 0] = 1e5
    add = torch.randn(1, H, W, C, generator=g)
    sc = spec.permute(0, 3, 1, 2, 4).double()
    comp = torch.complex(sc[..., 0].contiguous(), sc[..., 1].contiguous())
    # torch's c2r over the last axis after a full c2c over rows: the published irfftn (what the reference's FourierUnit calls)
    want = torch.fft.irfftn(comp, s=(H, W), dim=(-2, -1), norm="ortho").permute(0, 2, 3, 1) + add.double()
    t32 = torch.fft.irfftn(comp.to(torch.complex64), s=(H, W), dim=(-2, -1), norm="ortho").permute(0, 2, 3, 1).double() + add.double()
    y = torch.empty(1, H, W, C, device=gpu)
    tmp = torch.empty(1, H, Wf, 2 * C, device=gpu)
    ops.irfft2_f32(spec.reshape(1, H, Wf, 2 * C).to(gpu), tmp, y, add.to(gpu), 1, H, W, C, C, C,
                   lama._twiddles(W, gpu), lama._twiddles(H, gpu))
    e_hip, e_t = (y.cpu().double() - want).abs().max().item(), (t32 - want).abs().max().item()
    # the ignored imaginary parts must not leak: the result's scale is the 1e5 bin's, 2e5 / sqrt(HW)
    assert want.abs().max().item() < 4e5 / (H * W) ** 0.5 + 10
    assert e_hip <= max(6 * e_t, 0.1), (e_hip, e_t)                         # 0.1 = 1.6e-5 of the plane wave's amplitude


# ------------------------------------------------------------------ the generator on degenerate pictures
def _ocfg(cfg):
    from oracle import lama as olama
    return olama.LamaConfig(ngf=cfg.ngf, n_down=cfg.n_down, n_blocks=cfg.n_blocks, ratio_g=cfg.ratio_g, bn_eps=cfg.bn_eps)


@pytest.mark.parametrize("name", ["all_masked", "unmasked", "white_all_masked", "black_hole_in_white", "smallest_frame", "one_pixel_hole"])
def test_generator_on_degenerate_pictures_vs_oracle(gpu, name):
    from domain_rag_amd import lama
    from oracle import lama as olama
    cfg = lama.LamaConfig(ngf=16, n_blocks=2)
    p = lama.init_params(cfg, seed=9)
    net = lama.LamaHIP(cfg, p, gpu)
    rng = np.random.default_rng(3)
    H, W = (9, 9) if name == "smallest_frame" else (27, 41)                # 9 -> 16: the 2 x 2 bottleneck still reflects by 1
    img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    mask = np.zeros((H, W), np.uint8)
    if name == "all_masked":
        mask[:] = 255
    elif name == "white_all_masked":
        img[:] = 255; mask[:] = 1
    elif name == "black_hole_in_white":
        img[:] = 255; img[8:20, 10:30] = 0; mask[8:20, 10:30] = 200
    elif name == "smallest_frame":
        mask[2:6, 3:8] = 255
    elif name == "one_pixel_hole":
        mask[H - 1, W - 1] = 1                                             # the corner the symmetric padding mirrors three times
    out = net(torch.from_numpy(img).to(gpu), torch.from_numpy(mask).to(gpu)).cpu().numpy()
    ref = olama.inpaint(p, _ocfg(cfg), img, mask)
    ti, tm = olama.prepare_img_and_mask(img, mask)
    m = tm[0, 0].numpy() > 0
    assert out.shape == ref.shape
    assert np.array_equal(out[~m], ref[~m])
    if name == "unmasked":
        assert not m.any() and np.array_equal(out, np.pad(img, ((0, out.shape[0] - H), (0, out.shape[1] - W), (0, 0)), mode="symmetric"))
    pred_ref = olama.generator(p, _ocfg(cfg), torch.cat([ti * (1 - tm), tm], 1))
    pred = net._bufs["pred"][..., :3].cpu()
    assert (pred - pred_ref[0].permute(1, 2, 0)).abs().max().item() < 1e-3
    d = np.abs(out[m].astype(np.int32) - ref[m].astype(np.int32)) if m.any() else np.zeros(1, np.int32)
    assert d.max() <= 1 and (d > 0).mean() < 0.03
