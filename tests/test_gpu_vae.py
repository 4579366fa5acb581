"""Flux VAE on the HIP path (NHWC implicit-GEMM convs, GroupNorm+SiLU, mid attention) vs the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _rand(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).bfloat16()


@pytest.mark.parametrize("B,H,W,Ci,Co,stride", [(2, 8, 12, 64, 128, 1), (1, 16, 16, 128, 256, 1), (1, 10, 6, 256, 128, 1),
                                                (2, 8, 8, 128, 128, 2), (1, 33, 17, 64, 4, 1)])
def test_conv3x3(gpu, B, H, W, Ci, Co, stride):
    from domain_rag_amd import ops
    x = _rand((B, Ci, H, W), 1)
    w = _rand((Co, Ci, 3, 3), 2, 0.05)
    b = _rand((Co,), 3)
    if stride == 1:
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
        Ho, Wo, origin = H, W, 0
    else:
        ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
        Ho, Wo, origin = H // 2, W // 2, 1
    xp = torch.zeros((B, H + 2, W + 2, Ci), dtype=torch.bfloat16)
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    y = torch.empty((B, Ho, Wo, Co), dtype=torch.bfloat16, device=gpu)
    ops.conv3x3(xp.to(gpu), w.permute(0, 2, 3, 1).contiguous().to(gpu), y, B=B, Ho=Ho, Wo=Wo, Hp=H + 2, Wp=W + 2, Cin=Ci,
                Cout=Co, bias=b.to(gpu), stride=stride, oy=origin, ox=origin)
    assert _rel(y.permute(0, 3, 1, 2), ref) < 8e-3


def test_groupnorm_silu_and_pad(gpu):
    from domain_rag_amd import ops
    for C, H, W in [(128, 9, 7), (256, 40, 40), (512, 4, 4)]:
        B = 2
        x = _rand((B, C, H, W), C, 2.0) + 0.5
        g, b = _rand((C,), 1), _rand((C,), 2)
        ref = F.silu(F.group_norm(x, 32, g, b, 1e-6))
        y = torch.zeros((B, H + 2, W + 2, C), dtype=torch.bfloat16, device=gpu)
        ops.groupnorm_silu(x.permute(0, 2, 3, 1).contiguous().to(gpu), y, g.to(gpu), b.to(gpu), B, H, W, C, out_pad=1, silu=True)
        yc = y.cpu()
        assert _rel(yc[:, 1:-1, 1:-1].permute(0, 3, 1, 2), ref) < 1.5e-2
        assert (yc[:, 0] == 0).all() and (yc[:, -1] == 0).all() and (yc[:, :, 0] == 0).all() and (yc[:, :, -1] == 0).all()


def test_groupnorm_groups_with_a_large_mean_and_a_small_spread(gpu):
    """round 4: float32 partial sums of x and x^2 lose the variance of a group whose mean is large against its spread (E[x^2] - E[x]^2:
    1e4 * 1e-6 of rounding against a variance of 0.1); torch's GroupNorm (the reference's) does not.  The kernel sums x - c and (x - c)^2 with
    c = the group's first element.  Groups at mean 100 / -300 / 2000 with the smallest spread bf16 has there, next to ordinary groups, at the
    decode's largest map"""
    from domain_rag_amd import ops
    g_ = torch.Generator().manual_seed(9)
    for C, H, W in [(128, 96, 96), (256, 40, 40), (512, 16, 16)]:
        B = 2
        cpg = C // 32
        x = torch.randn(B, C, H, W, generator=g_)
        for grp, (mean, spread) in {1: (100.0, 0.5), 5: (-300.0, 2.0), 9: (2000.0, 16.0), 31: (64.0, 0.25)}.items():
            x[:, grp * cpg:(grp + 1) * cpg] = mean + spread * torch.randint(-1, 2, (B, cpg, H, W), generator=g_).float()
        x = x.bfloat16().float()
        g, b = _rand((C,), 1), _rand((C,), 2)
        ref = F.group_norm(x.double(), 32, g.double(), b.double(), 1e-6).float()
        y = torch.zeros((B, H, W, C), dtype=torch.bfloat16, device=gpu)
        ops.groupnorm_silu(x.permute(0, 2, 3, 1).contiguous().bfloat16().to(gpu), y, g.to(gpu), b.to(gpu), B, H, W, C, out_pad=0, silu=False)
        got = y.cpu().float().permute(0, 3, 1, 2)
        for grp in (1, 5, 9, 31, 0, 17):
            sl = slice(grp * cpg, (grp + 1) * cpg)
            assert _rel(got[:, sl], ref[:, sl]) < 1.5e-2, (C, grp, _rel(got[:, sl], ref[:, sl]))


def test_softmax_padcopy_pack(gpu):
    from domain_rag_amd import ops
    from oracle import vae as ov
    x = torch.randn(50, 256) * 5
    y = torch.empty((50, 256), dtype=torch.bfloat16, device=gpu)
    ops.softmax_rows(x.to(gpu), y, 50, 256, 0.3)
    assert _rel(y, torch.softmax(x * 0.3, -1)) < 1e-2
    x2 = torch.randn(7, 100) * 3          # ragged width: zero K-padding up to the row stride
    y2 = torch.full((7, 128), 9.0, dtype=torch.bfloat16, device=gpu)
    ops.softmax_rows(x2.to(gpu), y2, 7, 100, 1.0, ldy=128)
    assert _rel(y2[:, :100], torch.softmax(x2, -1)) < 1e-2 and (y2[:, 100:] == 0).all()
    a = _rand((2, 5, 3, 64), 4)
    up = torch.zeros((2, 12, 8, 64), dtype=torch.bfloat16, device=gpu)
    ops.pad_copy(a.to(gpu), up, 2, 5, 3, 64, upsample=2)
    ref = F.interpolate(a.permute(0, 3, 1, 2).float(), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(up.cpu()[:, 1:-1, 1:-1].float(), ref)
    # mask packing and latent unpack/pack round trip against the oracle layout
    mask = (torch.rand(2, 32, 48, generator=torch.Generator().manual_seed(0)) > 0.5).to(torch.uint8) * 255
    tok = torch.empty((2, 6, 256), dtype=torch.bfloat16, device=gpu)
    ops.mask_pack(mask.to(gpu), tok, 2, 32, 48, 256)
    assert torch.equal(tok.cpu().float(), ov.pack_mask(ov.preprocess_mask(mask)))
    t = _rand((2, 12, 64), 9)
    zp = torch.zeros((2, 8, 10, 64), dtype=torch.bfloat16, device=gpu)
    ops.unpack_latents(t.to(gpu), zp, 2, 3, 4, 64, 64, ov.SCALING, ov.SHIFT)
    ref = (ov.unpack_latents(t, 3, 4) / ov.SCALING + ov.SHIFT)
    # torch-CPU rounds the python scalar 0.1159 to bf16 before the add, CUDA/HIP keep it in fp32
    # (gpu_kernel_with_scalars opmath): <= 1 bf16 ulp apart
    got = zp.cpu()[:, 1:-1, 1:-1, :16].permute(0, 3, 1, 2).float()
    assert ((got - ref.float()).abs() <= ref.float().abs() * 2 ** -7 + 1e-3).all()


@pytest.mark.parametrize("blocks,layers,B,h,w", [((128, 256), 1, 2, 4, 4), ((128, 256, 512, 512), 2, 1, 4, 4),
                                                  ((128, 256), 1, 1, 5, 3)])
def test_vae_decode_encode_vs_oracle(gpu, blocks, layers, B, h, w):
    from domain_rag_amd import vae
    from oracle import vae as ov
    cfg = vae.VaeConfig(block_out_channels=blocks, layers_per_block=layers)
    p = vae.init_params(cfg, seed=3)
    model = vae.FluxVaeHIP(cfg, p, gpu)
    tok = _rand((B, h * w, 64), 5)
    img_u8, rows = model.decode_tokens(tok.to(gpu), B, h, w, return_rows=True)
    ref_u8, ref_img = ov.decode_tokens_to_u8(p, tok, h, w, block_out=blocks, layers=layers)
    p32 = {k: v.float() for k, v in p.items()}
    _, ref32 = ov.decode_tokens_to_u8(p32, tok.float(), h, w, block_out=blocks, layers=layers)
    H, W = img_u8.shape[1:3]
    got = (rows.view(B, H, W, -1)[..., :3].float().cpu() / 2 + 0.5).clamp(0, 1).permute(0, 3, 1, 2)
    e, e_or = _rel(got, ref32), _rel(ref_img, ref32)
    assert e < max(1e-2, 1.3 * e_or), f"decode: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"
    # pixels: within 1e-2 relative of full scale (= 2.55 LSB) of the fp32 oracle's pixels
    d = (img_u8.cpu().int() - (ref32.permute(0, 2, 3, 1) * 255).round().int()).abs()
    assert d.float().max().item() <= max(3.0, 255 * 1.3 * e_or), f"pixel levels {d.max().item()}, bf16 oracle vs f32 {e_or:.4e} of full scale (bar 1.3 x)"

    # encode (mode and sampled) -> packed tokens
    S = cfg.downscale
    Hi, Wi = 2 * h * S // 2 * 2, 2 * w * S // 2 * 2
    g = torch.Generator().manual_seed(7)
    img = (torch.rand(B, Hi, Wi, 3, generator=g) * 255).to(torch.uint8)
    mask = torch.zeros(B, Hi, Wi, dtype=torch.uint8); mask[:, : Hi // 2] = 255
    noise = torch.randn(B, 16, Hi // S, Wi // S, generator=g).bfloat16()
    for use_mask, nz in [(False, None), (True, noise)]:
        toks = torch.empty((B, (Hi // S // 2) * (Wi // S // 2), 64), dtype=torch.bfloat16, device=gpu)
        model.encode_to_tokens(img.to(gpu), mask.to(gpu) if use_mask else None, None if nz is None else nz.to(gpu), toks, 64)
        x = ov.preprocess_image(img)
        if use_mask:
            x = x * (1 - ov.preprocess_mask(mask))
        ref = ov.pack_latents(ov.sample_latents(ov.encode_moments(p32, x, block_out=blocks, layers=layers), None if nz is None else nz.float()))
        refb = ov.pack_latents(ov.sample_latents(ov.encode_moments(p, x.bfloat16(), block_out=blocks, layers=layers), nz))
        e, e_or = _rel(toks, ref), _rel(refb, ref)
        assert e < max(1.5e-2, 1.3 * e_or), f"encode mask={use_mask}: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"


def test_mid_attention_query_blocks_are_exact(gpu):
    """the mid-block attention bounds its fp32 score block by processing query rows in blocks: rows are independent, so
    any block size must give the same bits (block of 64 rows here vs one block)"""
    from domain_rag_amd import vae
    cfg = vae.VaeConfig(block_out_channels=(128, 256), layers_per_block=1)
    p = vae.init_params(cfg, seed=4)
    tok = _rand((2, 12 * 10, 64), 6).to(gpu)
    one = vae.FluxVaeHIP(cfg, p, gpu)
    ref = one.decode_tokens(tok, 2, 12, 10).clone()
    blocked = vae.FluxVaeHIP(cfg, p, gpu)
    blocked.attn_block_bytes = 6 * (24 * 20) * 64            # -> 64 query rows per block (T = 480)
    assert torch.equal(blocked.decode_tokens(tok, 2, 12, 10), ref)
