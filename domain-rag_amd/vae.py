"""Flux VAE (AutoencoderKL) on the HIP path: NHWC bf16, every 3x3 convolution an implicit GEMM on
the MFMA main loop, GroupNorm+SiLU fused and written straight into the zero-haloed buffer the next
convolution reads.  Mirrors ``AutoencoderKL.decode/encode`` as used at the end/start of
``pipe(...)`` / ``pipe_fill(...)`` (batch_generate_flux_kshot.py:467-474,
outpainting_updown_sampling_redux.py:1246-1257); parameter names are the diffusers state_dict names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import ops

SCALING, SHIFT = 0.3611, 0.1159


@dataclass
class VaeConfig:
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 16
    in_channels: int = 3

    @property
    def downscale(self) -> int:
        return 2 ** (len(self.block_out_channels) - 1)


def param_shapes(cfg: VaeConfig) -> dict[str, tuple]:
    s: dict[str, tuple] = {}
    bo = cfg.block_out_channels

    def conv(name, co, ci, k=3):
        s[name + ".weight"] = (co, ci, k, k)
        s[name + ".bias"] = (co,)

    def gn(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(pre, ci, co):
        gn(pre + "norm1", ci); conv(pre + "conv1", co, ci); gn(pre + "norm2", co); conv(pre + "conv2", co, co)
        if ci != co:
            conv(pre + "conv_shortcut", co, ci, 1)

    def mid(pre, c):
        resnet(pre + "resnets.0.", c, c)
        gn(pre + "attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[pre + f"attentions.0.{n}.weight"] = (c, c)
            s[pre + f"attentions.0.{n}.bias"] = (c,)
        resnet(pre + "resnets.1.", c, c)

    # encoder
    conv("encoder.conv_in", bo[0], cfg.in_channels)
    ci = bo[0]
    for i, co in enumerate(bo):
        for r in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{r}.", ci, co)
            ci = co
        if i < len(bo) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
    mid("encoder.mid_block.", bo[-1])
    gn("encoder.conv_norm_out", bo[-1])
    conv("encoder.conv_out", 2 * cfg.latent_channels, bo[-1])
    # decoder
    rev = tuple(reversed(bo))
    conv("decoder.conv_in", rev[0], cfg.latent_channels)
    mid("decoder.mid_block.", rev[0])
    ci = rev[0]
    for i, co in enumerate(rev):
        for r in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{r}.", ci, co)
            ci = co
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
    gn("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg.in_channels, rev[-1])
    return s


def init_params(cfg: VaeConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16) -> dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g, device=device) / math.sqrt(fan_in)
        out[name] = t.to(dtype)
    return out


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


class FluxVaeHIP:
    def __init__(self, cfg: VaeConfig, params: dict, device="cuda"):
        self.attn_block_bytes = 1 << 31        # bound of the mid-block attention's fp32 score block (see _attn)
        self.cfg, self.dev = cfg, torch.device(device)
        self.w: dict[str, torch.Tensor] = {}
        for k, v in params.items():
            v = v.to(self.dev, torch.bfloat16)
            if v.dim() == 4 and v.shape[-1] == 3:      # [Co, Ci, 3, 3] -> [Co4, 3, 3, Ci64]
                co, ci = v.shape[:2]
                w = torch.zeros((_pad4(co), 3, 3, _pad64(ci)), dtype=torch.bfloat16, device=self.dev)
                w[:co, :, :, :ci] = v.permute(0, 2, 3, 1)
                self.w[k] = w.contiguous()
            elif v.dim() == 4:                          # 1x1 shortcut
                self.w[k] = v[:, :, 0, 0].contiguous()
            elif k.endswith(".bias") and (k.endswith("conv_out.bias")):
                b = torch.zeros(_pad4(v.shape[0]), dtype=torch.bfloat16, device=self.dev)
                b[: v.shape[0]] = v
                self.w[k] = b
            else:
                self.w[k] = v.contiguous()
        self._bufs: dict = {}

    # ------------------------------------------------------------------ buffers
    def _buf(self, tag: str, shape, zero: bool = False, dtype=torch.bfloat16):
        key = (tag, tuple(shape), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
            self._bufs[key] = b
        return b

    def release(self):
        self._bufs.clear()

    def _frame(self, B, H, W):
        """work buffers are kept per (tag, shape); when the frame size changes the previous frame's set is dropped, so a
        long run over many image sizes does not accumulate one full buffer set per size"""
        key = (B, H, W)
        if getattr(self, "_frame_key", None) != key:
            self._bufs.clear()
            self._frame_key = key

    # ------------------------------------------------------------------ layers (NHWC)
    def _conv(self, xp, name, B, H, W, Cin, out_tag, resid=None, stride=1, origin=0, Ho=None, Wo=None):
        w = self.w[name + ".weight"]
        Cout = w.shape[0]
        Ho, Wo = Ho or H, Wo or W
        y = self._buf(out_tag, (B, Ho, Wo, Cout))
        ops.conv3x3(xp, w, y, B=B, Ho=Ho, Wo=Wo, Hp=H + 2, Wp=W + 2, Cin=_pad64(Cin), Cout=Cout,
                    bias=self.w[name + ".bias"], resid=resid, stride=stride, oy=origin, ox=origin)
        return y

    def _gn(self, x, name, B, H, W, C, silu=True, pad=1, tag="gn"):
        y = self._buf(f"{tag}{pad}", (B, H + 2 * pad, W + 2 * pad, C), zero=bool(pad))
        return ops.groupnorm_silu(x, y, self.w[name + ".weight"], self.w[name + ".bias"], B, H, W, C, out_pad=pad, silu=silu)

    def _resnet(self, x, pre, B, H, W, Ci, Co, tag):
        t = self._gn(x, pre + "norm1", B, H, W, Ci)
        h = self._conv(t, pre + "conv1", B, H, W, Ci, "res_h")
        t2 = self._gn(h, pre + "norm2", B, H, W, Co)
        if Ci != Co:
            sc = self._buf("res_sc", (B, H, W, Co))
            ops.gemm(x.view(-1, Ci), self.w[pre + "conv_shortcut.weight"], out=sc, bias=self.w[pre + "conv_shortcut.bias"])
        else:
            sc = x
        return self._conv(t2, pre + "conv2", B, H, W, Co, tag, resid=sc)

    def _attn(self, x, pre, B, H, W, C, tag):
        """single-head attention over H*W tokens (head_dim = C): GEMM / softmax / GEMM per image
        (head_dim 512 does not fit the fused MFMA attention kernel's register budget)."""
        n = self._gn(x, pre + "group_norm", B, H, W, C, silu=False, pad=0, tag="attn_n")
        T = H * W
        w = self.w
        q = self._buf("attn_q", (B * T, C)); k = self._buf("attn_k", (B * T, C))
        ops.gemm(n.view(-1, C), w[pre + "to_q.weight"], out=q, bias=w[pre + "to_q.bias"])
        ops.gemm(n.view(-1, C), w[pre + "to_k.weight"], out=k, bias=w[pre + "to_k.bias"])
        o = self._buf("attn_o", (B * T, C))
        Tp = (T + 63) // 64 * 64                       # K of the P·V GEMM, zero-padded
        vt = self._buf("attn_vt", (C, Tp), zero=True)
        # query rows are independent: the fp32 score block is bounded to ~2 GiB (T = 122 500 at the 2800-px cap would
        # otherwise need a 60 GB matrix); at 1024^2 (T = 16 384, 1.07 GB) it is still one block
        Tq = min(T, max(64, (self.attn_block_bytes // (6 * T)) // 64 * 64))
        s = self._buf("attn_s", (Tq, T), dtype=torch.float32)
        pbuf = self._buf("attn_p", (Tq, Tp))
        for b in range(B):
            nb = n.view(B, T, C)[b]
            # V^T = Wv · X^T (bias added after P·V: softmax rows sum to one)
            ops.gemm(w[pre + "to_v.weight"], nb, out=vt, M=C, lda=C, ldc=Tp)
            for r0 in range(0, T, Tq):
                rows = min(Tq, T - r0)
                ops.gemm(q.view(B, T, C)[b][r0:r0 + rows], k.view(B, T, C)[b], out=s, out_f32=True)
                ops.softmax_rows(s, pbuf, rows, T, 1.0 / math.sqrt(C), ldy=Tp)
                ops.gemm(pbuf, vt, out=o.view(B, T, C)[b][r0:r0 + rows], bias=w[pre + "to_v.bias"], M=rows, lda=Tp, ldc=C)
        y = self._buf(tag, (B, H, W, C))
        ops.gemm(o, w[pre + "to_out.0.weight"], out=y, bias=w[pre + "to_out.0.bias"], resid=x)
        return y

    def _mid(self, x, pre, B, H, W, C):
        x = self._resnet(x, pre + "resnets.0.", B, H, W, C, C, "mid_a")
        x = self._attn(x, pre + "attentions.0.", B, H, W, C, "mid_b")
        return self._resnet(x, pre + "resnets.1.", B, H, W, C, C, "mid_c")

    # ------------------------------------------------------------------ decode / encode
    def decode_tokens(self, tokens: torch.Tensor, B: int, h: int, w: int, ld: int = 64, return_rows: bool = False):
        """packed latents [B, h*w, ld] (first 64 columns) -> uint8 RGB [B, H, W, 3]"""
        cfg = self.cfg
        rev = tuple(reversed(cfg.block_out_channels))
        H, W = 2 * h, 2 * w
        self._frame(B, 16 * h, 16 * w)
        zp = self._buf("dec_z", (B, H + 2, W + 2, 64), zero=True)
        ops.unpack_latents(tokens, zp, B, h, w, ld, 64, SCALING, SHIFT)
        x = self._conv(zp, "decoder.conv_in", B, H, W, cfg.latent_channels, "dec_in")
        x = self._mid(x, "decoder.mid_block.", B, H, W, rev[0])
        ci = rev[0]
        flip = 0
        for i, co in enumerate(rev):
            for r in range(cfg.layers_per_block + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{r}.", B, H, W, ci, co, f"dec_x{flip}")
                flip ^= 1
                ci = co
            if i < len(rev) - 1:
                xp = self._buf("up_pad", (B, 2 * H + 2, 2 * W + 2, co), zero=True)
                ops.pad_copy(x, xp, B, H, W, co, upsample=2)
                H, W = 2 * H, 2 * W
                x = self._conv(xp, f"decoder.up_blocks.{i}.upsamplers.0.conv", B, H, W, co, f"dec_x{flip}")
                flip ^= 1
        t = self._gn(x, "decoder.conv_norm_out", B, H, W, ci)
        y = self._conv(t, "decoder.conv_out", B, H, W, ci, "dec_out")          # [B, H, W, 4]
        img = self._buf("dec_u8", (B, H, W, 3), dtype=torch.uint8)
        ops.image_postprocess(y, img, B * H * W, y.shape[-1])
        return (img, y) if return_rows else img

    def encode_to_tokens(self, img_u8: torch.Tensor, mask_u8, noise, tokens: torch.Tensor, ld: int):
        """uint8 RGB [B,H,W,3] (* (1 - mask) when given) -> sampled, shifted/scaled, packed latents written
        to ``tokens`` [B, (H/16)(W/16), ld] (64 columns).  ``noise``: bf16 [B,16,H/8,W/8] or None (mode)."""
        cfg = self.cfg
        bo = cfg.block_out_channels
        B, H, W, _ = img_u8.shape
        self._frame(B, H, W)
        xp = self._buf("enc_in", (B, H + 2, W + 2, 64), zero=True)
        ops.image_preprocess(img_u8, mask_u8, xp, B, H, W, 64)
        x = self._conv(xp, "encoder.conv_in", B, H, W, cfg.in_channels, "enc_x0")
        ci = bo[0]
        flip = 1
        for i, co in enumerate(bo):
            for r in range(cfg.layers_per_block):
                x = self._resnet(x, f"encoder.down_blocks.{i}.resnets.{r}.", B, H, W, ci, co, f"enc_x{flip}")
                flip ^= 1
                ci = co
            if i < len(bo) - 1:
                dp = self._buf("down_pad", (B, H + 2, W + 2, co), zero=True)
                ops.pad_copy(x, dp, B, H, W, co, upsample=1)
                x = self._conv(dp, f"encoder.down_blocks.{i}.downsamplers.0.conv", B, H, W, co, f"enc_x{flip}", stride=2,
                               origin=1, Ho=H // 2, Wo=W // 2)
                flip ^= 1
                H, W = H // 2, W // 2
        x = self._mid(x, "encoder.mid_block.", B, H, W, ci)
        t = self._gn(x, "encoder.conv_norm_out", B, H, W, ci)
        mom = self._conv(t, "encoder.conv_out", B, H, W, ci, "enc_mom")          # [B, h, w, 32]
        ops.sample_pack_latents(mom, noise, tokens, B, H, W, mom.shape[-1], ld, SCALING, SHIFT)
        return tokens, mom


def _pad4(c: int) -> int:
    return (c + 3) // 4 * 4
