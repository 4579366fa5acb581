"""CPU restatement of the LaMa inpainting stage (lama_inpaint/lama_inpaint.py:172-215 -> simple_lama_inpainting.SimpleLama).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for the network: ``simple-lama-inpainting`` (un-pinned
in requirements.txt, wraps the TorchScript export ``big-lama.pt`` of saic-mdal/lama's ``big-lama`` model) is neither
vendored under /root/reference nor installed here, and no weights exist on disk.  What follows restates the published
architecture (Suvorov et al., "Resolution-robust Large Mask Inpainting with Fourier Convolutions", WACV 2022;
``FFCResNetGenerator`` with the big-lama config: ngf 64, 3 down-samplings, 18 FFC residual blocks at global ratio 0.75,
no LFU, sigmoid output) in plain torch modules-as-functions: reflect-padded convs, eval-mode BatchNorm, torch.fft.
The mask construction IS pinned: tests/golden/make_lama_goldens.py captures it from the imported reference script.

Parameter naming follows the generator's nn.Sequential (``model.<i>.…``):
  1            FFC_BN_ACT(4 -> 64, k7) after ReflectionPad2d(3)                 (ffc.convl2l, bn_l)
  2, 3, 4      FFC_BN_ACT stride-2 3x3 down-samplings 64->128->256->512; the last one splits its output into
               local 128 | global 384 channels                                  (ffc.convl2l [+ ffc.convl2g], bn_l [+ bn_g])
  5 .. 22      FFCResnetBlock: conv1, conv2 = FFC_BN_ACT(512 -> 512, k3, reflect), x + conv2(conv1(x)) per branch
  24/27/30     ConvTranspose2d(k3, s2, p1, output_padding 1) 512->256->128->64, each + BatchNorm (25/28/31) + ReLU
  34           Conv2d(64 -> 3, k7) after ReflectionPad2d(3); 35 sigmoid
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class LamaConfig:
    ngf: int = 64
    n_down: int = 3
    n_blocks: int = 18
    ratio_g: float = 0.75
    bn_eps: float = 1e-5

    @property
    def dim(self):
        return self.ngf * 2 ** self.n_down

    @property
    def c_global(self):
        return int(self.dim * self.ratio_g)

    @property
    def c_local(self):
        return self.dim - self.c_global


def _bn(p, name, x, eps):
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                        training=False, eps=eps)


def _conv_reflect(x, w, stride=1):
    pad = w.shape[-1] // 2
    return F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, stride=stride)


def fourier_unit(p, pre, x, eps):
    """FourierUnit.forward (fft_norm='ortho'): rfft2 -> (re, im) stacked as channels 2c, 2c+1 -> 1x1 conv + BN + ReLU -> irfft2"""
    B, C, H, W = x.shape
    f = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")
    f = torch.stack((f.real, f.imag), dim=-1).permute(0, 1, 4, 2, 3).reshape(B, 2 * C, H, W // 2 + 1)
    f = F.relu(_bn(p, pre + ".bn", F.conv2d(f, p[pre + ".conv_layer.weight"]), eps))
    f = f.view(B, -1, 2, H, W // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()
    return torch.fft.irfftn(torch.complex(f[..., 0], f[..., 1]), s=(H, W), dim=(-2, -1), norm="ortho")


def spectral_transform(p, pre, x, eps):
    """SpectralTransform.forward (stride 1, enable_lfu False): conv2(conv1(x) + fu(conv1(x)))"""
    x = F.relu(_bn(p, pre + ".conv1.1", F.conv2d(x, p[pre + ".conv1.0.weight"]), eps))
    return F.conv2d(x + fourier_unit(p, pre + ".fu", x, eps), p[pre + ".conv2.weight"])


def ffc_bn_act(p, pre, xl, xg, eps, stride=1):
    """FFC_BN_ACT.forward: (x_l, x_g) -> (relu(bn_l(l2l(x_l) + g2l(x_g))), relu(bn_g(l2g(x_l) + g2g(x_g))));
    a branch whose channel count is zero is absent (its conv is an Identity that contributes 0)."""
    ol = _conv_reflect(xl, p[pre + ".ffc.convl2l.weight"], stride)
    if xg is not None:
        ol = ol + _conv_reflect(xg, p[pre + ".ffc.convg2l.weight"], stride)
    ol = F.relu(_bn(p, pre + ".bn_l", ol, eps))
    og = None
    if pre + ".ffc.convl2g.weight" in p:
        og = _conv_reflect(xl, p[pre + ".ffc.convl2g.weight"], stride)
        if xg is not None:
            og = og + spectral_transform(p, pre + ".ffc.convg2g", xg, eps)
        og = F.relu(_bn(p, pre + ".bn_g", og, eps))
    return ol, og


def generator(p: dict, cfg: LamaConfig, x: torch.Tensor) -> torch.Tensor:
    """FFCResNetGenerator.forward on [B, 4, H, W] (masked image | mask), H and W multiples of 8 -> [B, 3, H, W] in (0, 1)"""
    eps = cfg.bn_eps
    xl = F.relu(_bn(p, "model.1.bn_l", F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), p["model.1.ffc.convl2l.weight"]), eps))
    xg = None
    i = 2
    for _ in range(cfg.n_down):
        xl, xg = ffc_bn_act(p, f"model.{i}", xl, xg, eps, stride=2)
        i += 1
    for _ in range(cfg.n_blocks):
        yl, yg = ffc_bn_act(p, f"model.{i}.conv1", xl, xg, eps)
        yl, yg = ffc_bn_act(p, f"model.{i}.conv2", yl, yg, eps)
        xl, xg = xl + yl, xg + yg
        i += 1
    x = torch.cat([xl, xg], dim=1)
    i += 1                                               # ConcatTupleLayer
    for _ in range(cfg.n_down):
        x = F.conv_transpose2d(x, p[f"model.{i}.weight"], p[f"model.{i}.bias"], stride=2, padding=1, output_padding=1)
        x = F.relu(_bn(p, f"model.{i + 1}", x, eps))
        i += 3
    i += 1                                               # ReflectionPad2d(3)
    x = F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), p[f"model.{i}.weight"], p[f"model.{i}.bias"])
    return torch.sigmoid(x)


def prepare_img_and_mask(image_u8: np.ndarray, mask_u8: np.ndarray, modulo: int = 8):
    """simple_lama_inpainting.utils.prepare_img_and_mask: /255 float32 CHW, np.pad(..., mode='symmetric') of the bottom /
    right edges up to a multiple of 8, mask binarised with ``> 0`` AFTER the padding."""
    img = np.transpose(image_u8.astype(np.float32) / 255, (2, 0, 1))
    msk = mask_u8.astype(np.float32)[None] / 255
    H, W = img.shape[1:]
    ph, pw = (-H) % modulo, (-W) % modulo
    img = np.pad(img, ((0, 0), (0, ph), (0, pw)), mode="symmetric")
    msk = np.pad(msk, ((0, 0), (0, ph), (0, pw)), mode="symmetric")
    return torch.from_numpy(img)[None], (torch.from_numpy(msk)[None] > 0).float()


def inpaint(p: dict, cfg: LamaConfig, image_u8: np.ndarray, mask_u8: np.ndarray) -> np.ndarray:
    """SimpleLama.__call__: [H,W,3] uint8 + [H,W] uint8 mask (non-zero = fill) -> uint8 [ceil8(H), ceil8(W), 3].
    The exported module computes generator(cat(img * (1 - m), m)) and blends m * predicted + (1 - m) * img
    (DefaultInpaintingTrainingModule.forward, concat_mask=True); the wrapper scales by 255, clips and TRUNCATES to uint8
    and returns the padded frame as it is."""
    img, m = prepare_img_and_mask(image_u8, mask_u8)
    pred = generator(p, cfg, torch.cat([img * (1 - m), m], dim=1))
    out = m * pred + (1 - m) * img
    return np.clip(out[0].permute(1, 2, 0).numpy() * 255, 0, 255).astype(np.uint8)
