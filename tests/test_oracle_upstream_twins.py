"""Pins of the generation-core oracle against UPSTREAM code that is importable in this image.

diffusers 0.33.1 — the wheel that holds the reference's arithmetic — is not installable here, but `transformers` ships modules of
the same lineage with the same arithmetic.  Each test below builds the upstream module, shares random weights with the oracle's
functional restatement and compares the outputs.  What each twin covers (and what stays review-only) is tabulated in
oracle/__init__.py.

  oracle/vae.py   resnet / mid_attention / up-, down-sample / encode_moments / decode
                      <- transformers.models.janus.modeling_janus.JanusVQVAE{ResnetBlock, AttnBlock, ConvUpsample, ConvDownsample,
                         Encoder, Decoder} (the LDM `Encoder` / `Decoder` that AutoencoderKL also descends from)
  oracle/flux.py  rms_norm            <- transformers T5LayerNorm (fp32 variance, cast to the weight dtype, times weight)
                  apply_rope          <- GPT-J apply_rotary_pos_emb + rotate_every_two (interleaved pairs)
                  rope_tables         <- GPT-J create_sinusoidal_positions (per axis)
                  adaln_zero, gated_mlp_residual, adaln_continuous
                                      <- qwen2_5_omni Qwen2_5_OmniAdaLayerNormZero, DiTDecoderLayer, ..._Final (F5-TTS lineage:
                                         diffusers' AdaLayerNormZero / AdaLayerNormContinuous chunk orders)
"""
import math

import pytest
import torch

tf = pytest.importorskip("transformers")

from oracle import flux as oflux  # noqa: E402
from oracle import vae as ovae  # noqa: E402


# ------------------------------------------------------------------------------------------------ VAE (Janus VQVAE twins)
def _janus():
    from transformers.models.janus import modeling_janus as mj
    return mj


def _janus_cfg(base=32, mult=(1, 2, 2), nres=1, latent=4, double=True):
    mj = _janus()
    return mj.JanusVQVAEConfig(base_channels=base, channel_multiplier=list(mult), num_res_blocks=nres, latent_channels=latent,
                               double_latent=double, in_channels=3, out_channels=3, dropout=0.0)


def _randomize(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in sorted(mod.named_parameters()):
            if name.endswith("weight") and prm.dim() == 1:      # norm scales around 1
                prm.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))
            elif prm.dim() == 1:
                prm.copy_(0.1 * torch.randn(prm.shape, generator=g))
            else:
                fan_in = prm[0].numel()
                prm.copy_(torch.randn(prm.shape, generator=g) / math.sqrt(fan_in))
    return mod.eval()


def _resnet_params(block, pre):
    """JanusVQVAEResnetBlock -> the diffusers ResnetBlock2D names oracle.vae.resnet reads (`nin_shortcut` is diffusers' 1x1
    `conv_shortcut`)"""
    p = {}
    for n in ("norm1", "conv1", "norm2", "conv2"):
        p[pre + n + ".weight"] = getattr(block, n).weight.detach().clone()
        p[pre + n + ".bias"] = getattr(block, n).bias.detach().clone()
    if block.in_channels != block.out_channels:
        p[pre + "conv_shortcut.weight"] = block.nin_shortcut.weight.detach().clone()
        p[pre + "conv_shortcut.bias"] = block.nin_shortcut.bias.detach().clone()
    return p


def _attn_params(block, pre):
    """JanusVQVAEAttnBlock's 1x1 convolutions -> the Linear weights of diffusers' Attention (to_q / to_k / to_v / to_out.0)"""
    p = {pre + "group_norm.weight": block.norm.weight.detach().clone(), pre + "group_norm.bias": block.norm.bias.detach().clone()}
    for src, dst in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")):
        p[pre + dst + ".weight"] = getattr(block, src).weight.detach()[:, :, 0, 0].clone()
        p[pre + dst + ".bias"] = getattr(block, src).bias.detach().clone()
    return p


def _close(a, b, rel=2e-5):
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 64), (64, 32)])
def test_vae_resnet_block_equals_upstream_twin(cin, cout):
    mj = _janus()
    blk = _randomize(mj.JanusVQVAEResnetBlock(_janus_cfg(), cin, cout), 10 + cin + 2 * cout)
    p = _resnet_params(blk, "r.")
    assert (blk.norm1.num_groups, blk.norm1.eps, blk.conv1.padding) == (32, 1e-6, (1, 1))
    g = torch.Generator().manual_seed(1)
    for shape in ((2, cin, 16, 16), (1, cin, 9, 13)):
        x = torch.randn(shape, generator=g)
        with torch.no_grad():
            _close(ovae.resnet(x, p, "r."), blk(x.clone()))


def test_vae_mid_attention_equals_upstream_twin():
    mj = _janus()
    blk = _randomize(mj.JanusVQVAEAttnBlock(64), 3)
    p = _attn_params(blk, "a.")
    g = torch.Generator().manual_seed(2)
    for shape in ((2, 64, 8, 8), (1, 64, 5, 11)):
        x = torch.randn(shape, generator=g) * 2.0
        with torch.no_grad():
            _close(ovae.mid_attention(x, p, "a."), blk(x.clone()))


def test_vae_up_and_down_sample_equal_upstream_twins():
    """decode's `interpolate(nearest, 2x) -> conv 3x3 pad 1` and encode_moments' `pad (0, 1, 0, 1) -> conv 3x3 stride 2 pad 0`"""
    import torch.nn.functional as F
    mj = _janus()
    up = _randomize(mj.JanusVQVAEConvUpsample(32), 4)
    dn = _randomize(mj.JanusVQVAEConvDownsample(32), 5)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 10, 14, generator=g)
    with torch.no_grad():
        pu = {"u.weight": up.conv.weight, "u.bias": up.conv.bias}
        assert torch.equal(ovae._conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), pu, "u"), up(x))
        pd = {"d.weight": dn.conv.weight, "d.bias": dn.conv.bias}
        for xx in (x, x[:, :, :9, :13]):                                          # even and odd sizes
            assert torch.equal(ovae._conv(F.pad(xx, (0, 1, 0, 1)), pd, "d", stride=2, padding=0), dn(xx))


def _silence_level_attention(levels):
    """The LDM stacks of Janus carry an AttnBlock behind every ResNet block of their LOWEST-resolution level, AutoencoderKL's
    stacks do not.  With `proj_out` zeroed such a block returns `residual + 0`: the identity, bit for bit."""
    with torch.no_grad():
        for lvl in levels:
            for a in lvl.attn:
                a.proj_out.weight.zero_()
                a.proj_out.bias.zero_()


def _mid_params(mid, pre):
    p = {}
    p.update(_resnet_params(mid.block_1, pre + "resnets.0."))
    p.update(_attn_params(mid.attn_1, pre + "attentions.0."))
    p.update(_resnet_params(mid.block_2, pre + "resnets.1."))
    return p


def _conv_params(conv, name):
    return {name + ".weight": conv.weight.detach().clone(), name + ".bias": conv.bias.detach().clone()}


def decoder_params(dec):
    """JanusVQVAEDecoder -> diffusers `decoder.*` names (up level j of Janus' list = up_blocks.j)"""
    p = _conv_params(dec.conv_in, "decoder.conv_in")
    p.update(_mid_params(dec.mid, "decoder.mid_block."))
    for j, lvl in enumerate(dec.up):
        for r, blk in enumerate(lvl.block):
            p.update(_resnet_params(blk, f"decoder.up_blocks.{j}.resnets.{r}."))
        if hasattr(lvl, "upsample"):
            p.update(_conv_params(lvl.upsample.conv, f"decoder.up_blocks.{j}.upsamplers.0.conv"))
    p.update(_conv_params(dec.norm_out, "decoder.conv_norm_out"))
    p.update(_conv_params(dec.conv_out, "decoder.conv_out"))
    return p


def encoder_params(enc):
    p = _conv_params(enc.conv_in, "encoder.conv_in")
    for i, lvl in enumerate(enc.down):
        for r, blk in enumerate(lvl.block):
            p.update(_resnet_params(blk, f"encoder.down_blocks.{i}.resnets.{r}."))
        if hasattr(lvl, "downsample"):
            p.update(_conv_params(lvl.downsample.conv, f"encoder.down_blocks.{i}.downsamplers.0.conv"))
    p.update(_mid_params(enc.mid, "encoder.mid_block."))
    p.update(_conv_params(enc.norm_out, "encoder.conv_norm_out"))
    p.update(_conv_params(enc.conv_out, "encoder.conv_out"))
    return p


@pytest.mark.parametrize("mult,nres", [((1, 2, 2), 1), ((1, 2, 4, 4), 2)])
def test_vae_decoder_stack_equals_upstream_twin(mult, nres):
    """conv_in -> mid (resnet, attention, resnet) -> per level (nres + 1) resnets [+ nearest 2x upsample + conv, all levels but
    the last] -> GroupNorm -> swish -> conv_out: the order, the channel plan (block_out reversed, first resnet of a level takes
    the previous level's width) and where the up-samplers sit.  (1, 2, 4, 4) x 2 is the Flux VAE's own plan at base 32."""
    mj = _janus()
    cfg = _janus_cfg(base=32, mult=mult, nres=nres, latent=16)
    dec = _randomize(mj.JanusVQVAEDecoder(cfg), 7)
    _silence_level_attention([dec.up[0]])
    p = decoder_params(dec)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(2, 16, 6, 5, generator=g)
    with torch.no_grad():
        want = dec(z.clone())
        got = ovae.decode(p, z, block_out=tuple(32 * m for m in mult), layers=nres)
    assert got.shape == want.shape == (2, 3, 6 * 2 ** (len(mult) - 1), 5 * 2 ** (len(mult) - 1))
    _close(got, want, rel=5e-5)


@pytest.mark.parametrize("mult,nres", [((1, 2, 2), 1), ((1, 2, 4, 4), 2)])
def test_vae_encoder_stack_equals_upstream_twin(mult, nres):
    """conv_in -> per level nres resnets [+ (0, 1, 0, 1) pad + stride-2 conv, all but the last] -> mid -> GroupNorm -> swish ->
    conv_out to 2 x latent channels (mean | logvar)"""
    mj = _janus()
    cfg = _janus_cfg(base=32, mult=mult, nres=nres, latent=16, double=True)
    enc = _randomize(mj.JanusVQVAEEncoder(cfg), 8)
    _silence_level_attention([enc.down[-1]])
    p = encoder_params(enc)
    g = torch.Generator().manual_seed(5)
    f = 2 ** (len(mult) - 1)
    x = torch.randn(2, 3, 5 * f, 6 * f, generator=g)
    with torch.no_grad():
        want = enc(x.clone())
        got = ovae.encode_moments(p, x, block_out=tuple(32 * m for m in mult), layers=nres)
    assert got.shape == want.shape == (2, 32, 5, 6)
    _close(got, want, rel=5e-5)


# ------------------------------------------------------------------------------------------------ RMSNorm (T5LayerNorm twin)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rms_norm_equals_t5_layer_norm_bit_for_bit(dtype):
    from transformers.models.t5.modeling_t5 import T5LayerNorm
    g = torch.Generator().manual_seed(6)
    m = T5LayerNorm(128, eps=1e-6)
    with torch.no_grad():
        m.weight.copy_(torch.rand(128, generator=g) + 0.5)
    m = m.to(dtype)
    x = (torch.randn(2, 24, 37, 128, generator=g) * torch.logspace(-3, 3, 37)[None, None, :, None]).to(dtype)
    with torch.no_grad():
        want = m(x)
    got = oflux.rms_norm(x, m.weight.detach())
    assert got.dtype == want.dtype == dtype
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ RoPE (GPT-J twin)
def test_apply_rope_equals_gptj_rotary_bit_for_bit():
    """Flux rotates interleaved pairs (x0, x1), (x2, x3), ... — GPT-J's `rotate_every_two` convention.  GPT-J's layout is
    [B, S, H, D] with sin / cos [B, S, D/2]; the oracle's [B, H, S, D] with [S, D/2].  The oracle computes in float32 and casts back
    (diffusers' apply_rotary_emb): the float32 case is bit-identical to GPT-J, the bf16 case to GPT-J run on the float copy."""
    from transformers.models.gptj.modeling_gptj import apply_rotary_pos_emb
    g = torch.Generator().manual_seed(7)
    B, H, S, D = 2, 3, 29, 128
    ang = torch.rand(S, D // 2, generator=g) * 50.0
    cos, sin = ang.cos(), ang.sin()
    x = torch.randn(B, H, S, D, generator=g)
    want = apply_rotary_pos_emb(x.transpose(1, 2), sin[None].expand(B, -1, -1), cos[None].expand(B, -1, -1)).transpose(1, 2)
    assert torch.equal(oflux.apply_rope(x, cos, sin), want)
    xb = x.bfloat16()
    want_b = apply_rotary_pos_emb(xb.float().transpose(1, 2), sin[None].expand(B, -1, -1), cos[None].expand(B, -1, -1)).transpose(1, 2)
    got_b = oflux.apply_rope(xb, cos, sin)
    assert got_b.dtype == torch.bfloat16 and torch.equal(got_b, want_b.bfloat16())


def test_rope_tables_equal_gptj_sinusoid_per_axis():
    """per axis: angle = position * 10000^(-2i/d) — GPT-J's create_sinusoidal_positions (float32; the oracle's float64 table is
    held to float32 resolution of the angle), concatenated over Flux's axes (16, 56, 56)"""
    from transformers.models.gptj.modeling_gptj import create_sinusoidal_positions
    axes = (16, 56, 56)
    npos = 96
    ids = torch.stack([torch.zeros(npos), torch.arange(npos).float(), torch.arange(npos).flip(0).float()], dim=1)
    cos, sin = oflux.rope_tables(ids, axes)
    assert cos.shape == sin.shape == (npos, 64)
    off = 0
    for a, d in enumerate(axes):
        tab = create_sinusoidal_positions(npos, d)                    # [pos, d/2 sin | d/2 cos]
        rows = ids[:, a].long()
        s_ref, c_ref = tab[rows, : d // 2], tab[rows, d // 2:]
        # |d(sin)| <= |d(angle)|: float32 angle of up to 95 rad carries ~4e-6 of error
        assert (sin[:, off:off + d // 2] - s_ref).abs().max() < 2e-5
        assert (cos[:, off:off + d // 2] - c_ref).abs().max() < 2e-5
        off += d // 2


# ------------------------------------------------------------------------------------------------ AdaLN-Zero (Qwen2.5-Omni DiT twins)
def _omni():
    from transformers.models.qwen2_5_omni import modeling_qwen2_5_omni as mq
    return mq


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adaln_zero_chunk_order_equals_upstream_twin(dtype):
    mq = _omni()
    D = 64
    m = _randomize(mq.Qwen2_5_OmniAdaLayerNormZero(D), 9).to(dtype)
    p = {"n.weight": m.linear.weight.detach(), "n.bias": m.linear.bias.detach()}
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 11, D, generator=g).to(dtype)
    temb = torch.randn(2, D, generator=g).to(dtype)
    with torch.no_grad():
        want = m(x, emb=temb)
    got = oflux.adaln_zero(p, "n", temb, x)
    assert len(got) == len(want) == 5
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adaln_continuous_scale_shift_order_equals_upstream_twin(dtype):
    mq = _omni()
    D = 64
    m = _randomize(mq.Qwen2_5_OmniAdaLayerNormZero_Final(D), 10).to(dtype)
    p = {"n.weight": m.linear.weight.detach(), "n.bias": m.linear.bias.detach()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 11, D, generator=g).to(dtype)
    temb = torch.randn(2, D, generator=g).to(dtype)
    with torch.no_grad():
        want = m(x, temb)
    assert torch.equal(oflux.adaln_continuous(p, "n", temb, x), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gated_residual_and_mlp_half_of_a_block_equals_upstream_twin(dtype):
    """DiTDecoderLayer with its attention replaced by a fixed Linear (the attention itself is not Flux's): what is pinned is
    x + gate_msa * attn, LN * (1 + scale_mlp) + shift_mlp, Linear -> GELU(tanh) -> Linear, x + gate_mlp * ff — the image (and,
    with the context names, text) stream of a Flux double block around its joint attention."""
    mq = _omni()
    D = 64
    cfg = mq.Qwen2_5OmniDiTConfig(hidden_size=D, num_attention_heads=2, head_dim=32, ff_mult=4, dropout=0.0)
    layer = mq.DiTDecoderLayer(cfg)

    class FixedAttention(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(D, D)

        def forward(self, hidden_states, position_embeddings=None, attention_mask=None):
            return self.lin(hidden_states)

    layer.attn = FixedAttention()
    layer = _randomize(layer, 11).to(dtype)
    p = {"n.weight": layer.attn_norm.linear.weight.detach(), "n.bias": layer.attn_norm.linear.bias.detach(),
         "ff.net.0.proj.weight": layer.ff.ff[0].weight.detach(), "ff.net.0.proj.bias": layer.ff.ff[0].bias.detach(),
         "ff.net.2.weight": layer.ff.ff[3].weight.detach(), "ff.net.2.bias": layer.ff.ff[3].bias.detach()}
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 13, D, generator=g).to(dtype)
    temb = torch.randn(2, D, generator=g).to(dtype)
    with torch.no_grad():
        want = layer(x, temb, position_embeddings=None, block_diff=torch.zeros(1))
        n, g_msa, sh_mlp, sc_mlp, g_mlp = oflux.adaln_zero(p, "n", temb, x)
        got = oflux.gated_mlp_residual(p, "ff.", x, layer.attn.lin(n), g_msa, sh_mlp, sc_mlp, g_mlp)
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ packing (torch's own pixel (un)shuffle)
def test_latent_and_mask_packing_equal_torch_pixel_unshuffle():
    """FluxPipeline._pack_latents / _unpack_latents and FluxFillPipeline.prepare_mask_latents are view / permute / reshape chains in diffusers;
    torch ships the same index maps as operators of its own: pack = pixel_unshuffle(2) flattened to tokens (channel c * 4 + dy * 2 + dx),
    unpack = pixel_shuffle(2), the mask's 8 x 8 fold = pixel_unshuffle(8) on the single mask channel"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 16, 12, 20, generator=g)
    tok = ovae.pack_latents(z)
    assert tok.shape == (2, 6 * 10, 64)
    assert torch.equal(tok, F.pixel_unshuffle(z, 2).flatten(2).transpose(1, 2))
    assert torch.equal(ovae.unpack_latents(tok, 6, 10), z)
    assert torch.equal(ovae.unpack_latents(tok, 6, 10), F.pixel_shuffle(tok.transpose(1, 2).reshape(2, 64, 6, 10), 2))
    m = (torch.rand(2, 1, 32, 48, generator=g) > 0.5).float()
    pm = ovae.pack_mask(m)
    assert pm.shape == (2, 2 * 3, 256)
    assert torch.equal(pm, ovae.pack_latents(F.pixel_unshuffle(m, 8)))
