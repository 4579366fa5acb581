"""Level-2 drop-in boundary: objects with the names, constructor chains and call signatures the
reference scripts use, backed by the HIP path.  ``install()`` registers them as the modules the scripts import,
so the call sites work unchanged:

    clip.load("ViT-B/32", device) -> (model, preprocess); model.encode_image(x)        retrieval/…:209,171
    faiss.IndexFlatIP(d); .add(x); .search(q, k) -> (D, I)                              retrieval/…:425-434
    FluxPriorReduxPipeline.from_pretrained(dir, text_encoder=…, …, torch_dtype).to(dev)
        (image=[PIL…], prompt=…, prompt_2=…, prompt_embeds_scale=[…], pooled_prompt_embeds_scale=[…])
        -> mapping-and-attribute object                                                 batch_…:139-146,459-465
    FluxPipeline.from_pretrained(dir, torch_dtype).to(dev)(guidance_scale, num_inference_steps, height, width,
        generator, prompt_embeds, pooled_prompt_embeds).images                          batch_…:148-151,467-474
    FluxFillPipeline.from_pretrained(dir, …).to(dev)(image, mask_image, height, width, guidance_scale,
        num_inference_steps, prompt_embeds, pooled_prompt_embeds, generator, strength).images[0]
                                                                                        outpainting_…:534-541,1246-1257

There are no checkpoints offline: ``DRAG_SYNTHETIC_WEIGHTS=1`` builds seeded random weights of the same
architectures and ``DRAG_TINY=1`` the reduced test configuration; without them ``from_pretrained`` reads the
checkpoint directories the reference reads and raises if they are missing.  Text encoders / tokenizers passed to
``FluxPriorReduxPipeline.from_pretrained`` (the reference passes its ``transformers`` modules) encode a prompt the
first time it is seen; the result is cached on disk (``TextCache``, engine.py) — the prompt is constant per dataset.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from . import engine as E, redux as redux_mod, vae as vae_mod, vit as vit_mod
from .fill_pipeline import FluxFillHIP
from .flux import FluxTransformerHIP
from .flux_params import FluxConfig, init_params, load_safetensors_dir


def _flags():
    return os.environ.get("DRAG_SYNTHETIC_WEIGHTS", "0") == "1", os.environ.get("DRAG_TINY", "0") == "1"


class PriorOutput(dict):
    """FluxPriorReduxPipelineOutput: unpacks as ``**mapping`` (batch_…:473) and exposes attributes (outpainting_…:1253)"""
    __getattr__ = dict.__getitem__


class PipelineOutput:
    def __init__(self, images):
        self.images = images


class _Pipe:
    _seed = 0

    def __init__(self, path, kwargs):
        self._path, self._kwargs, self._built, self.device = path, kwargs, False, None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        return cls(str(pretrained_model_name_or_path), kwargs)

    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("domain_rag_amd pipelines run on the MI355X HIP path only (device must be 'cuda')")
        if not self._built:
            self.device = dev
            self._build(dev)
            self._built = True
        return self

    def _require(self):
        if not self._built:
            raise RuntimeError("call .to('cuda') first")


class FluxPriorReduxPipeline(_Pipe):
    def _build(self, dev):
        synthetic, tiny = _flags()
        vitcfg = vit_mod.VitConfig(**E.TINY["vit"]) if tiny else vit_mod.VitConfig.siglip_so400m()
        J = E.TINY["flux"]["joint_attention_dim"] if tiny else 4096
        P = E.TINY["flux"]["pooled_projection_dim"] if tiny else 768
        if synthetic:
            vitp = vit_mod.init_generic_params(vitcfg, self._seed + 2, device=dev)
            rp = redux_mod.init_redux_params(vitcfg.hidden, J, seed=self._seed + 3, device=dev)
        else:
            if not os.path.isdir(self._path):
                raise FileNotFoundError(f"{self._path} not found (set DRAG_SYNTHETIC_WEIGHTS=1 for seeded random weights)")
            vitp = vit_mod.siglip_to_generic(load_safetensors_dir(os.path.join(self._path, "image_encoder")), vitcfg)
            rp = load_safetensors_dir(os.path.join(self._path, "image_embedder"))
        self.vit_cfg = vitcfg
        self.prior = redux_mod.ReduxPriorHIP(vitcfg, vitp, rp, dev)
        kw = self._kwargs
        self.text = E.TextCache(os.path.dirname(self._path.rstrip("/")) or ".", synthetic,
                                E.TINY["t5_tokens"] if tiny else redux_mod.T5_TOKENS, J, P, dev,
                                encoders=(kw.get("text_encoder"), kw.get("text_encoder_2"), kw.get("tokenizer"), kw.get("tokenizer_2")))

    def __call__(self, image, prompt="", prompt_2="", prompt_embeds_scale=1.0, pooled_prompt_embeds_scale=1.0, **_):
        self._require()
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        n = len(images)

        def per_image(v):
            v = list(v) if isinstance(v, (list, tuple)) else [v] * n
            if len(v) != n:
                raise ValueError(f"number of scales ({len(v)}) must match number of images ({n})")
            return [float(x) for x in v]

        def one(p):
            if isinstance(p, (list, tuple)):
                if len(set(p)) > 1:
                    raise NotImplementedError("per-image prompts differ; the reference always passes one constant prompt")
                return p[0] if p else ""
            return p or ""

        t5, pooled = self.text.get(one(prompt), one(prompt_2))
        x = E.siglip_input_device(images, self.vit_cfg.image_size, self.device)
        pe, pp = self.prior(x, t5, pooled, per_image(prompt_embeds_scale), per_image(pooled_prompt_embeds_scale), group=n)
        return PriorOutput(prompt_embeds=pe, pooled_prompt_embeds=pp)


def _draws(generator, seed_default, B, H, W, n):
    """n sequential bf16 [B,16,H/8,W/8] draws from the caller's CPU generator (diffusers ``randn_tensor``)"""
    if generator is None:
        generator = torch.Generator("cpu").manual_seed(seed_default)
    if isinstance(generator, (list, tuple)):
        raise NotImplementedError("one generator per call (the reference passes one)")
    if generator.device.type != "cpu":
        raise NotImplementedError("the reference seeds a CPU generator (batch_…:468, outpainting_…:1231)")
    return [torch.randn((B, 16, H // 8, W // 8), generator=generator, dtype=torch.bfloat16) for _ in range(n)]


def _step_hook(pipe, callback):
    """diffusers' ``callback_on_step_end(pipe, step_index, timestep, {"latents": packed latents}) -> dict``: observers only
    (the returned dict is ignored; the reference passes no callback)"""
    if callback is None:
        return None
    return lambda i, latents: callback(pipe, i, None, {"latents": latents})


class _FluxPipe(_Pipe):
    _kind = "dev"

    def _build(self, dev):
        synthetic, tiny = _flags()
        fkw = dict(E.TINY["flux"]) if tiny else {}
        cfg = FluxConfig(in_channels=384 if self._kind == "fill" else 64, **fkw)
        vcfg = vae_mod.VaeConfig(**(E.TINY["vae"] if tiny else {}))
        if synthetic:
            tp = init_params(cfg, seed=self._seed, device=dev)
            vp = vae_mod.init_params(vcfg, seed=self._seed + 1, device=dev)
        else:
            if not os.path.isdir(self._path):
                raise FileNotFoundError(f"{self._path} not found (set DRAG_SYNTHETIC_WEIGHTS=1 for seeded random weights)")
            cfg = FluxConfig.from_json(os.path.join(self._path, "transformer", "config.json"))
            tp = load_safetensors_dir(os.path.join(self._path, "transformer"))
            vp = load_safetensors_dir(os.path.join(self._path, "vae"))
        self.cfg = cfg
        self.tr, self.vae = FluxTransformerHIP(cfg, tp, dev), vae_mod.FluxVaeHIP(vcfg, vp, dev)

    @staticmethod
    def _to_pil(out_u8):
        from PIL import Image
        return [Image.fromarray(a) for a in out_u8.cpu().numpy()]


class FluxPipeline(_FluxPipe):
    _kind = "dev"

    def _build(self, dev):
        super()._build(dev)
        self.pipe = E.FluxTxt2ImgHIP(self.tr, self.vae)

    def __call__(self, prompt=None, guidance_scale=3.5, num_inference_steps=28, height=1024, width=1024, generator=None,
                 prompt_embeds=None, pooled_prompt_embeds=None, callback_on_step_end=None, **_):
        self._require()
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("prompt_embeds and pooled_prompt_embeds are required (the reference always passes the Redux prior's)")
        B = prompt_embeds.shape[0]
        height, width = height // 16 * 16, width // 16 * 16
        noise = E.pack_noise(_draws(generator, 0, B, height, width, 1)[0])
        out = self.pipe(prompt_embeds, pooled_prompt_embeds, height=height, width=width, guidance_scale=guidance_scale,
                        num_inference_steps=num_inference_steps, noise_tokens=noise, on_step=_step_hook(self, callback_on_step_end))
        return PipelineOutput(self._to_pil(out))


class FluxFillPipeline(_FluxPipe):
    _kind = "fill"

    def _build(self, dev):
        super()._build(dev)
        self.pipe = FluxFillHIP(self.tr, self.vae)

    def __call__(self, prompt=None, image=None, mask_image=None, height=1024, width=1024, guidance_scale=30.0,
                 num_inference_steps=50, prompt_embeds=None, pooled_prompt_embeds=None, generator=None, strength=1.0,
                 callback_on_step_end=None, **_):
        from PIL import Image
        self._require()
        if image is None or mask_image is None:
            raise ValueError("image and mask_image are required")
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise ValueError("prompt_embeds and pooled_prompt_embeds are required (the reference always passes the Redux prior's)")
        if not 0.0 <= strength <= 1.0:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if int(num_inference_steps * strength) < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                             "pipeline steps is 0 which is < 1 and not appropriate for this pipeline.")
        W16, H16 = max(width // 16 * 16, 16), max(height // 16 * 16, 16)
        # VaeImageProcessor.preprocess: resize to (width, height) with LANCZOS when the sizes differ
        im = image.convert("RGB")
        mk = mask_image.convert("L")
        if im.size != (W16, H16):
            im = im.resize((W16, H16), Image.LANCZOS)
        if mk.size != (W16, H16):
            mk = mk.resize((W16, H16), Image.LANCZOS)
        img_u8 = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())[None].to(self.device)
        msk_u8 = torch.from_numpy(np.asarray(mk, dtype=np.uint8).copy())[None].to(self.device)
        enc_n, noise, menc_n = _draws(generator, 0, 1, H16, W16, 3)       # draw order of FluxFillPipeline.__call__
        out = self.pipe(img_u8, msk_u8, prompt_embeds, pooled_prompt_embeds, guidance_scale=guidance_scale,
                        num_inference_steps=num_inference_steps, strength=strength, enc_noise=enc_n.to(self.device),
                        masked_enc_noise=menc_n.to(self.device), noise_tokens=E.pack_noise(noise).to(self.device),
                        on_step=_step_hook(self, callback_on_step_end))
        return PipelineOutput(self._to_pil(out))


def _clip_module():
    from . import retrieval as R
    m = types.ModuleType("clip")

    def load(name="ViT-B/32", device="cuda", **_):
        synthetic, _tiny = _flags()
        weights = None if synthetic else os.environ.get("DRAG_CLIP_WEIGHTS")
        if not synthetic and not weights:
            raise FileNotFoundError("set DRAG_CLIP_WEIGHTS to a ViT-B/32 state dict (.pt/.safetensors) or DRAG_SYNTHETIC_WEIGHTS=1")
        return R.load_clip(name, device=device, weights=weights)
    m.load = load
    m.available_models = lambda: ["ViT-B/32"]
    return m


def _faiss_module():
    from . import retrieval as R
    m = types.ModuleType("faiss")
    m.IndexFlatIP = R.IndexFlatIP
    return m


def _diffusers_module():
    m = types.ModuleType("diffusers")
    m.FluxPriorReduxPipeline, m.FluxPipeline, m.FluxFillPipeline = FluxPriorReduxPipeline, FluxPipeline, FluxFillPipeline
    return m


def _simple_lama_module():
    from . import lama
    mod = types.ModuleType("simple_lama_inpainting")
    mod.SimpleLama = lama.SimpleLama
    return mod


def install(modules=("clip", "faiss", "diffusers", "simple_lama_inpainting")) -> None:
    """register the HIP-backed stand-ins under the module names the reference scripts import"""
    makers = {"clip": _clip_module, "faiss": _faiss_module, "diffusers": _diffusers_module, "simple_lama_inpainting": _simple_lama_module}
    for name in modules:
        sys.modules[name] = makers[name]()
