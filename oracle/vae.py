"""oracle/vae.py — CPU restatement (plain torch, NCHW) of the Flux VAE (AutoencoderKL) and the
pipeline glue around it.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates diffusers 0.33.1 ``AutoencoderKL.encode/decode`` (Encoder/Decoder/UNetMidBlock2D/
ResnetBlock2D/Attention/Upsample2D/Downsample2D), ``DiagonalGaussianDistribution``,
``FluxPipeline._pack_latents/_unpack_latents``, ``FluxFillPipeline.prepare_mask_latents`` and
``VaeImageProcessor`` — un-vendored; anchored on pipe(...) batch_generate_flux_kshot.py:467-474 and
pipe_fill(...) outpainting_updown_sampling_redux.py:1246-1257.  Not pinned on the reference (it holds no tests); the blocks and both stacks ARE pinned on their importable upstream twins
(transformers JanusVQVAE*, tests/test_oracle_upstream_twins.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SCALING, SHIFT = 0.3611, 0.1159


def _conv(x, p, name, stride=1, padding=1):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def _gn(x, p, name):
    return F.group_norm(x, 32, p[name + ".weight"], p[name + ".bias"], 1e-6)


def resnet(x, p, pre):
    h = _conv(F.silu(_gn(x, p, pre + "norm1")), p, pre + "conv1")
    h = _conv(F.silu(_gn(h, p, pre + "norm2")), p, pre + "conv2")
    if pre + "conv_shortcut.weight" in p:
        x = _conv(x, p, pre + "conv_shortcut", padding=0)
    return x + h


def mid_attention(x, p, pre):
    B, C, H, W = x.shape
    res = x
    h = _gn(x.view(B, C, H * W), p, pre + "group_norm").transpose(1, 2)  # [B, HW, C]
    q = F.linear(h, p[pre + "to_q.weight"], p[pre + "to_q.bias"])
    k = F.linear(h, p[pre + "to_k.weight"], p[pre + "to_k.bias"])
    v = F.linear(h, p[pre + "to_v.weight"], p[pre + "to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, p[pre + "to_out.0.weight"], p[pre + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + res


def decode(p, z, block_out=(128, 256, 512, 512), layers=2):
    x = _conv(z, p, "decoder.conv_in")
    x = resnet(x, p, "decoder.mid_block.resnets.0.")
    x = mid_attention(x, p, "decoder.mid_block.attentions.0.")
    x = resnet(x, p, "decoder.mid_block.resnets.1.")
    n = len(block_out)
    for i in range(n):
        for r in range(layers + 1):
            x = resnet(x, p, f"decoder.up_blocks.{i}.resnets.{r}.")
        if i < n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(x, p, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    x = F.silu(_gn(x, p, "decoder.conv_norm_out"))
    return _conv(x, p, "decoder.conv_out")


def encode_moments(p, x, block_out=(128, 256, 512, 512), layers=2):
    x = _conv(x, p, "encoder.conv_in")
    n = len(block_out)
    for i in range(n):
        for r in range(layers):
            x = resnet(x, p, f"encoder.down_blocks.{i}.resnets.{r}.")
        if i < n - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = _conv(x, p, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    x = resnet(x, p, "encoder.mid_block.resnets.0.")
    x = mid_attention(x, p, "encoder.mid_block.attentions.0.")
    x = resnet(x, p, "encoder.mid_block.resnets.1.")
    x = F.silu(_gn(x, p, "encoder.conv_norm_out"))
    return _conv(x, p, "encoder.conv_out")


def sample_latents(moments, noise=None):
    """DiagonalGaussianDistribution(moments).sample() (noise given) or .mode(); then (z - shift) * scaling"""
    mean, logvar = moments.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return (z - SHIFT) * SCALING


def pack_latents(z):
    B, C, H, W = z.shape
    z = z.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return z.reshape(B, (H // 2) * (W // 2), C * 4)


def unpack_latents(tok, h, w):
    """tok [B, h*w, 64] -> [B, 16, 2h, 2w]"""
    B, _, ch = tok.shape
    z = tok.view(B, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return z.reshape(B, ch // 4, 2 * h, 2 * w)


def decode_tokens_to_u8(p, tok, h, w, **kw):
    """pipeline tail: unpack -> /scaling + shift -> decode -> postprocess to uint8 HWC"""
    z = unpack_latents(tok, h, w)
    z = z / SCALING + SHIFT
    img = decode(p, z, **kw)
    img = (img / 2 + 0.5).clamp(0, 1)
    arr = img.permute(0, 2, 3, 1).float().numpy()
    return torch.from_numpy((arr * 255).round().astype("uint8")), img


def preprocess_image(img_u8):
    """VaeImageProcessor.preprocess: uint8 HWC -> float32 NCHW in [-1, 1]"""
    x = img_u8.float() / 255.0
    return (2.0 * x - 1.0).permute(0, 3, 1, 2)


def preprocess_mask(mask_u8):
    """mask_processor (binarize, grayscale, no normalize): [B,H,W] uint8 -> float {0,1} [B,1,H,W]"""
    m = mask_u8.float() / 255.0
    return (m >= 0.5).float()[:, None]


def pack_mask(mask01):
    """prepare_mask_latents: [B,1,H,W] -> 8x8 pixel-unshuffle -> [B,64,H/8,W/8] -> _pack_latents -> [B, n, 256]"""
    B, _, H, W = mask01.shape
    m = mask01[:, 0].view(B, H // 8, 8, W // 8, 8).permute(0, 2, 4, 1, 3).reshape(B, 64, H // 8, W // 8)
    return pack_latents(m)
