"""LaMa inpainting generator on the HIP path — what ``simple_lama_inpainting.SimpleLama`` is to
lama_inpaint/lama_inpaint.py:104,172-215: ``result = simple_lama(image, mask)`` -> PIL image.

The wheel (un-pinned in requirements.txt) wraps the TorchScript export ``big-lama.pt`` of the big-lama
FFCResNetGenerator (ngf 64, 3 down-samplings, 18 FFC residual blocks with 75 % global channels, no LFU, sigmoid
output); prepare_img_and_mask pads bottom/right to a multiple of 8 (symmetric), the exported module blends
``mask * predicted + (1 - mask) * image`` and the wrapper returns the padded frame, x255, clipped, truncated to uint8.
Everything runs in float32 like the reference: convolutions as NHWC implicit GEMMs on the f32 matrix core
(csrc/lama.hip), the FourierUnit transforms by direct summation.  No CPU path.
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass

import numpy as np
import torch

from . import ops


@dataclass
class LamaConfig:
    ngf: int = 64
    n_down: int = 3
    n_blocks: int = 18
    ratio_g: float = 0.75
    bn_eps: float = 1e-5

    @property
    def dim(self) -> int:
        return self.ngf * 2 ** self.n_down

    @property
    def c_global(self) -> int:
        return int(self.dim * self.ratio_g)

    @property
    def c_local(self) -> int:
        return self.dim - self.c_global


def init_params(cfg: LamaConfig, seed: int = 0) -> dict:
    """seeded synthetic state dict with the generator's parameter names (``model.<i>.…``), float32, on the CPU"""
    g = torch.Generator().manual_seed(seed)
    p: dict = {}

    def conv(name, cout, cin, k, gain=1.0):
        p[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (gain * math.sqrt(2.0 / (cin * k * k)))

    def bn(name, c):
        p[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        p[name + ".bias"] = 0.05 * torch.randn(c, generator=g)
        p[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        p[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)

    conv("model.1.ffc.convl2l", cfg.ngf, 4, 7); bn("model.1.bn_l", cfg.ngf)
    i, c = 2, cfg.ngf
    for d in range(cfg.n_down):
        if d < cfg.n_down - 1:
            conv(f"model.{i}.ffc.convl2l", 2 * c, c, 3); bn(f"model.{i}.bn_l", 2 * c)
        else:
            conv(f"model.{i}.ffc.convl2l", cfg.c_local, c, 3); conv(f"model.{i}.ffc.convl2g", cfg.c_global, c, 3)
            bn(f"model.{i}.bn_l", cfg.c_local); bn(f"model.{i}.bn_g", cfg.c_global)
        c *= 2
        i += 1
    cl, cg = cfg.c_local, cfg.c_global
    for _ in range(cfg.n_blocks):
        for cv in ("conv1", "conv2"):
            pre = f"model.{i}.{cv}"
            conv(pre + ".ffc.convl2l", cl, cl, 3, 0.5); conv(pre + ".ffc.convl2g", cg, cl, 3, 0.5); conv(pre + ".ffc.convg2l", cl, cg, 3, 0.5)
            st = pre + ".ffc.convg2g"
            conv(st + ".conv1.0", cg // 2, cg, 1); bn(st + ".conv1.1", cg // 2)
            conv(st + ".fu.conv_layer", cg, cg, 1); bn(st + ".fu.bn", cg)
            conv(st + ".conv2", cg, cg // 2, 1, 0.5)
            bn(pre + ".bn_l", cl); bn(pre + ".bn_g", cg)
        i += 1
    i += 1
    for _ in range(cfg.n_down):
        p[f"model.{i}.weight"] = torch.randn(c, c // 2, 3, 3, generator=g) * math.sqrt(2.0 / (c * 9 / 4))
        p[f"model.{i}.bias"] = 0.05 * torch.randn(c // 2, generator=g)
        bn(f"model.{i + 1}", c // 2)
        c //= 2
        i += 3
    i += 1
    conv(f"model.{i}", 3, c, 7)
    p[f"model.{i}.bias"] = 0.05 * torch.randn(3, generator=g)
    return p


_KEY = re.compile(r"(?:^|\.)(model\.\d+\..*)$")


def normalize_state_dict(sd: dict) -> dict:
    """strip whatever wraps the generator in an export (``generator.``, ``model.generator.`` …) down to ``model.<i>.…``;
    BatchNorm ``num_batches_tracked`` counters are dropped"""
    out = {}
    for k, v in sd.items():
        m = _KEY.search(k)
        if m is None or k.endswith("num_batches_tracked"):
            continue
        key = m.group(1)
        # a wrapper attribute that is itself called "model" in front of the Sequential: keep the innermost match
        while True:
            m2 = _KEY.search(key[len("model."):])
            if m2 is None:
                break
            key = m2.group(1)
        out[key] = v.detach().float().cpu()
    return out


def config_from_state_dict(sd: dict) -> LamaConfig:
    ngf = sd["model.1.ffc.convl2l.weight"].shape[0]
    idx = sorted({int(k.split(".")[1]) for k in sd})
    blocks = [i for i in idx if f"model.{i}.conv1.ffc.convl2l.weight" in sd]
    n_down = blocks[0] - 2
    cl = sd[f"model.{blocks[0]}.conv1.ffc.convl2l.weight"].shape[0]
    dim = ngf * 2 ** n_down
    return LamaConfig(ngf=ngf, n_down=n_down, n_blocks=len(blocks), ratio_g=1.0 - cl / dim)


def load_state_dict(path: str) -> dict:
    """``big-lama.pt`` (TorchScript, what simple-lama downloads to ~/.cache/torch/hub/checkpoints and LAMA_MODEL overrides)
    or a plain checkpoint with ``state_dict`` / ``generator.*`` keys"""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj.state_dict()
    sd = normalize_state_dict(sd)
    if "model.1.ffc.convl2l.weight" not in sd:
        raise RuntimeError(f"{path}: no FFCResNetGenerator parameters (model.<i>.…) found")
    return sd


def _twiddles(n: int, device) -> torch.Tensor:
    a = 2.0 * np.pi * np.arange(n, dtype=np.float64) / n
    return torch.from_numpy(np.stack([np.cos(a), np.sin(a)], axis=1).astype(np.float32)).to(device)


class LamaHIP:
    """the generator with device-resident, kernel-layout weights; call it like SimpleLama: (image, mask) -> uint8 frame"""

    def __init__(self, cfg: LamaConfig, params: dict, device="cuda"):
        self.cfg, self.dev = cfg, torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("LamaHIP needs a GPU device (domain-rag_amd has no CPU path)")
        self._tw: dict = {}
        self._bufs: dict = {}
        self._frame = None
        p = params
        dev = self.dev

        def cw(name):                                    # nn.Conv2d [Cout, Cin, KH, KW] -> [Cout, KH, KW, Cin]
            return p[name + ".weight"].float().permute(0, 2, 3, 1).contiguous()

        def bn(name, bias=None):
            s = p[name + ".weight"].float() / torch.sqrt(p[name + ".running_var"].float() + cfg.bn_eps)
            t = p[name + ".bias"].float() - p[name + ".running_mean"].float() * s
            if bias is not None:
                t = t + bias.float() * s
            return s, t

        def up(*ts):
            return tuple(t.contiguous().to(dev) for t in ts)

        self.init = up(cw("model.1.ffc.convl2l"), *bn("model.1.bn_l"))
        self.down = []
        i = 2
        for d in range(cfg.n_down):
            if d < cfg.n_down - 1:
                self.down.append(up(cw(f"model.{i}.ffc.convl2l"), *bn(f"model.{i}.bn_l")))
            else:
                sl, tl = bn(f"model.{i}.bn_l"); sg, tg = bn(f"model.{i}.bn_g")
                self.down.append(up(torch.cat([cw(f"model.{i}.ffc.convl2l"), cw(f"model.{i}.ffc.convl2g")], 0), torch.cat([sl, sg]), torch.cat([tl, tg])))
            i += 1
        self.blocks = []
        for _ in range(cfg.n_blocks):
            blk = []
            for cv in ("conv1", "conv2"):
                pre = f"model.{i}.{cv}"
                st = pre + ".ffc.convg2g"
                blk.append(dict(
                    w_l=up(torch.cat([cw(pre + ".ffc.convl2l"), cw(pre + ".ffc.convg2l")], 3))[0], bn_l=up(*bn(pre + ".bn_l")),
                    w_g=up(cw(pre + ".ffc.convl2g"))[0], bn_g=up(*bn(pre + ".bn_g")),
                    st1=up(cw(st + ".conv1.0"), *bn(st + ".conv1.1")), fu=up(cw(st + ".fu.conv_layer"), *bn(st + ".fu.bn")),
                    st2=up(cw(st + ".conv2"))[0]))
            self.blocks.append(blk)
            i += 1
        i += 1
        self.ups = []
        for _ in range(cfg.n_down):
            wt = p[f"model.{i}.weight"].float().permute(1, 2, 3, 0).contiguous()     # ConvTranspose2d [Cin, Cout, KH, KW]
            self.ups.append(up(wt, *bn(f"model.{i + 1}", p[f"model.{i}.bias"])))
            i += 3
        i += 1
        self.final = up(cw(f"model.{i}"), p[f"model.{i}.bias"].float())

    @classmethod
    def from_file(cls, path: str, device="cuda") -> "LamaHIP":
        sd = load_state_dict(path)
        return cls(config_from_state_dict(sd), sd, device)

    def _buf(self, name, *shape, dtype=torch.float32):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != shape:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self._bufs[name] = t
        return t

    def _twiddle(self, n: int) -> torch.Tensor:
        t = self._tw.get(n)
        if t is None:
            if len(self._tw) > 64:
                self._tw.clear()
            t = self._tw[n] = _twiddles(n, self.dev)
        return t

    # ---- one FFC_BN_ACT of a residual block: X [h,w,dim] (local | global) -> Y, optional residual --------------------
    def _ffc(self, X, Y, q, h, w, resid):
        cfg = self.cfg
        cl, cg, D = cfg.c_local, cfg.c_global, cfg.dim
        half, wf = cg // 2, w // 2 + 1
        t1, t3 = self._buf("t1", h, w, half), self._buf("t3", h, w, half)
        f, f2, tmp = self._buf("f", h, wf, cg), self._buf("f2", h, wf, cg), self._buf("ftmp", h, wf, cg)
        s = self._buf("s", h, w, cg)
        tw_w, tw_h = self._twiddle(w), self._twiddle(h)
        xg = X.view(-1)[cl:]
        geo = dict(B=1, Hi=h, Wi=w, Ho=h, Wo=w)
        relu = ops.CONV_ACT_RELU
        # SpectralTransform: conv1 (1x1 + BN + ReLU) -> x + FourierUnit(x) -> conv2 (1x1)
        ops.conv2d_f32(xg, q["st1"][0], t1, Cin=cg, ldx=D, ldy=half, scale=q["st1"][1], shift=q["st1"][2], act=relu, **geo)
        ops.rfft2_f32(t1, tmp, f, 1, h, w, half, half, tw_w, tw_h)
        ops.conv2d_f32(f, q["fu"][0], f2, B=1, Hi=h, Wi=wf, Ho=h, Wo=wf, Cin=cg, ldx=cg, ldy=cg, scale=q["fu"][1], shift=q["fu"][2], act=relu)
        ops.irfft2_f32(f2, tmp, t3, t1, 1, h, w, half, half, half, tw_w, tw_h)
        ops.conv2d_f32(t3, q["st2"], s, Cin=half, ldx=half, ldy=cg, **geo)
        # local out = BN_l(l2l(x_l) + g2l(x_g)); global out = BN_g(l2g(x_l) + spectral(x_g)); 3x3, reflect
        rl = None if resid is None else resid
        rg = None if resid is None else resid.view(-1)[cl:]
        ops.conv2d_f32(X, q["w_l"], Y, Cin=D, ldx=D, ldy=D, pad=1, pad_mode=ops.PAD_REFLECT, scale=q["bn_l"][0], shift=q["bn_l"][1],
                       act=relu, resid=rl, ld_res=D, **geo)
        ops.conv2d_f32(X, q["w_g"], Y.view(-1)[cl:], Cin=cl, ldx=D, ldy=D, pad=1, pad_mode=ops.PAD_REFLECT, scale=q["bn_g"][0],
                       shift=q["bn_g"][1], act=relu, addend=s, ld_add=cg, resid=rg, ld_res=D, **geo)

    def generator_on_frame(self, x0: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
        """x0 NHWC f32 [Hp, Wp, 4] (masked image | mask), Hp and Wp multiples of 2**n_down -> NHWC f32 [Hp, Wp, 4] (3 used)"""
        cfg = self.cfg
        if Hp % (1 << cfg.n_down) or Wp % (1 << cfg.n_down):
            raise ValueError(f"LaMa frame {Hp}x{Wp} is not a multiple of {1 << cfg.n_down}")
        if self._frame != (Hp, Wp):                       # one frame's buffers at a time
            self._bufs.clear()
            self._frame = (Hp, Wp)
        relu, refl = ops.CONV_ACT_RELU, ops.PAD_REFLECT
        c = cfg.ngf
        a = self._buf("a0", Hp, Wp, c)
        ops.conv2d_f32(x0, self.init[0], a, B=1, Hi=Hp, Wi=Wp, Ho=Hp, Wo=Wp, Cin=4, ldx=4, ldy=c, pad=3, pad_mode=refl,
                       scale=self.init[1], shift=self.init[2], act=relu)
        h, w = Hp, Wp
        for d, (wd, sd, td) in enumerate(self.down):
            co = wd.shape[0]
            b = self._buf(f"d{d}", h // 2, w // 2, co)
            ops.conv2d_f32(a, wd, b, B=1, Hi=h, Wi=w, Ho=h // 2, Wo=w // 2, Cin=c, ldx=c, ldy=co, stride=2, pad=1, pad_mode=refl,
                           scale=sd, shift=td, act=relu)
            a, c, h, w = b, co, h // 2, w // 2
        X, Y, Z = a, self._buf("rb1", h, w, c), self._buf("rb2", h, w, c)
        for blk in self.blocks:
            self._ffc(X, Y, blk[0], h, w, None)
            self._ffc(Y, Z, blk[1], h, w, X)
            X, Z = Z, X
        a = X
        for u, (wu, su, tu) in enumerate(self.ups):
            co = wu.shape[0]
            b = self._buf(f"u{u}", 2 * h, 2 * w, co)
            ops.conv2d_f32(a, wu, b, B=1, Hi=h, Wi=w, Ho=2 * h, Wo=2 * w, Cin=c, ldx=c, ldy=co, stride=2, pad=1, transposed=True,
                           scale=su, shift=tu, act=relu)
            a, c, h, w = b, co, 2 * h, 2 * w
        pred = self._buf("pred", Hp, Wp, 4)
        ops.conv2d_f32(a, self.final[0], pred, B=1, Hi=h, Wi=w, Ho=h, Wo=w, Cin=c, ldx=c, ldy=4, pad=3, pad_mode=refl,
                       shift=self.final[1], act=ops.CONV_ACT_SIGMOID)
        return pred

    def __call__(self, image_u8: torch.Tensor, mask_u8: torch.Tensor) -> torch.Tensor:
        """uint8 RGB [H,W,3] + uint8 mask [H,W] (non-zero = fill), both on the device -> uint8 [ceil8(H), ceil8(W), 3]"""
        H, W, _ = image_u8.shape
        if tuple(mask_u8.shape) != (H, W):
            raise ValueError(f"mask {tuple(mask_u8.shape)} does not match the image {(H, W)}")
        mod = 1 << self.cfg.n_down
        Hp, Wp = -(-H // mod) * mod, -(-W // mod) * mod
        image_u8, mask_u8 = image_u8.contiguous(), mask_u8.contiguous()
        if self._frame != (Hp, Wp):
            self._bufs.clear()
            self._frame = (Hp, Wp)
        x0 = self._buf("x0", Hp, Wp, 4)
        ops.lama_prepare(image_u8, mask_u8, x0, H, W, Hp, Wp)
        pred = self.generator_on_frame(x0, Hp, Wp)
        out = torch.empty((Hp, Wp, 3), dtype=torch.uint8, device=self.dev)
        ops.lama_blend(pred, 4, image_u8, mask_u8, out, H, W, Hp, Wp)
        return out


class SimpleLama:
    """drop-in for ``simple_lama_inpainting.SimpleLama`` (lama_inpaint/lama_inpaint.py:5,104,174): ``SimpleLama()(image, mask)``
    with PIL inputs -> PIL image.  Weights: $LAMA_MODEL (the wheel's own override) or ./model/big-lama.pt; with
    DRAG_SYNTHETIC_WEIGHTS=1 a seeded random generator of the same architecture (tests, benchmarks)."""

    def __init__(self, device=None):
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if os.environ.get("DRAG_SYNTHETIC_WEIGHTS"):
            cfg = LamaConfig(ngf=16, n_blocks=2) if os.environ.get("DRAG_TINY") else LamaConfig()
            self.model = LamaHIP(cfg, init_params(cfg, seed=11), dev)
        else:
            path = os.environ.get("LAMA_MODEL") or os.path.join(".", "model", "big-lama.pt")
            if not os.path.exists(path):
                raise FileNotFoundError(f"LaMa weights not found at {path} (set LAMA_MODEL to big-lama.pt)")
            self.model = LamaHIP.from_file(path, dev)
        self.device = dev

    def __call__(self, image, mask):
        from PIL import Image
        img = np.array(image)
        msk = np.array(mask)
        if img.ndim != 3 or img.shape[2] != 3:
            # the reference falls into its RuntimeError("expected input ... channels") branch for non-RGB input
            raise RuntimeError(f"expected input with 3 channels, got an image array of shape {img.shape}")
        if msk.ndim == 3:
            msk = msk[:, :, 0]
        out = self.model(torch.from_numpy(np.ascontiguousarray(img)).to(self.device),
                         torch.from_numpy(np.ascontiguousarray(msk)).to(self.device))
        return Image.fromarray(out.cpu().numpy())
