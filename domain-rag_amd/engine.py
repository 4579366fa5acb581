"""Model assembly for the stage CLIs: load the FLUX.1 checkpoints the reference loads
(``load_model``: batch_generate_flux_kshot.py:117-153, outpainting_updown_sampling_redux.py:500-543) into the
HIP classes, or build seeded synthetic weights of the same architectures when no checkpoints exist
(``synthetic=True``; there are none offline).  Models are loaded ONCE per process (the reference reloads them per
sample at outpainting_…:1185 — a pure slowdown).

Text encoders: the prompt is constant per dataset ("" everywhere except FISH, outpainting_…:85-95), so T5 /
CLIP-text outputs are cached inputs (``prompt_embeds_cache``), never on the per-image path.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch

from . import ops, redux as redux_mod, vae as vae_mod, vit as vit_mod
from .fill_pipeline import FluxFillHIP
from .flux import FluxTransformerHIP, latent_image_ids
from .flux_params import FluxConfig, init_params, load_safetensors_dir
from .scheduler import flow_sigmas, model_timestep

TINY = dict(flux=dict(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=256, pooled_projection_dim=64),
            vae=dict(layers_per_block=1), vit=dict(image_size=56, patch_size=14, hidden=192, heads=2, layers=1, intermediate=304),
            t5_tokens=16)


def pack_noise(noise_nchw: torch.Tensor) -> torch.Tensor:
    """FluxPipeline._pack_latents on a host noise tensor [B,16,h,w] -> [B,(h/2)(w/2),64] (layout only)"""
    B, C, H, W = noise_nchw.shape
    return noise_nchw.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 2) * (W // 2), C * 4).contiguous()


def generator_noise(seed: int, B: int, H: int, W: int, n_draws: int):
    """the draws diffusers takes from ``torch.Generator("cpu").manual_seed(seed)``: bf16 [B,16,H/8,W/8] each, in order"""
    g = torch.Generator("cpu").manual_seed(int(seed))
    return [torch.randn((B, 16, H // 8, W // 8), generator=g, dtype=torch.bfloat16) for _ in range(n_draws)]


def siglip_input(pil_images, size: int) -> torch.Tensor:
    """SiglipImageProcessor: RGB, resize to (size,size) bicubic (aspect ignored); rescale/normalise happen on device"""
    from PIL import Image
    arr = [np.asarray(im.convert("RGB").resize((size, size), Image.BICUBIC), dtype=np.uint8) for im in pil_images]
    return torch.from_numpy(np.stack(arr))


def siglip_input_device(pil_images, size: int, device) -> torch.Tensor:
    """same as ``siglip_input`` with the BICUBIC resize on the GPU (``drag_resample_u8``, bit-identical to PIL):
    raw decoded bytes are uploaded once, nothing is resized on the host -> uint8 [n, size, size, 3] on device"""
    from . import resample
    out = torch.empty((len(pil_images), size, size, 3), dtype=torch.uint8, device=device)
    for i, im in enumerate(pil_images):
        raw = torch.from_numpy(np.array(im.convert("RGB"), dtype=np.uint8, copy=True)).to(device)
        resample.siglip_resize_u8(raw, size, out=out[i])
    return out


def encode_prompt_with(text_encoder, text_encoder_2, tokenizer, tokenizer_2, prompt: str, prompt_2: str = "", max_sequence_length: int = 512):
    """FluxPriorReduxPipeline.encode_prompt with the caller's own ``transformers`` modules (the ones the reference builds
    in ``load_model``, batch_…:120-137 / outpainting_…:505-522): pooled = CLIP-L ``pooler_output`` of ``prompt`` (77 tokens,
    padded / truncated), embeds = T5 last hidden state of ``prompt_2 or prompt`` (512 tokens).  Runs once per distinct
    prompt (the prompt is constant per dataset) and is cached; it is not on the per-image path."""
    prompt_2 = prompt_2 or prompt
    with torch.no_grad():
        ids = tokenizer([prompt], padding="max_length", max_length=getattr(tokenizer, "model_max_length", 77), truncation=True,
                        return_tensors="pt").input_ids
        pooled = text_encoder(ids.to(next(text_encoder.parameters()).device), output_hidden_states=False).pooler_output
        ids2 = tokenizer_2([prompt_2], padding="max_length", max_length=max_sequence_length, truncation=True,
                           return_tensors="pt").input_ids
        embeds = text_encoder_2(ids2.to(next(text_encoder_2.parameters()).device), output_hidden_states=False)[0]
    return embeds[0].float().cpu(), pooled[0].float().cpu()


def load_text_encoders(flux_dir: str, device="cuda"):
    """the four objects the reference's ``load_model`` builds from ``<model>/FLUX.1-*/{text_encoder, text_encoder_2,
    tokenizer, tokenizer_2}`` (batch_…:120-137), bf16"""
    from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5TokenizerFast
    te = CLIPTextModel.from_pretrained(flux_dir, subfolder="text_encoder", torch_dtype=torch.bfloat16).to(device).eval()
    te2 = T5EncoderModel.from_pretrained(flux_dir, subfolder="text_encoder_2", torch_dtype=torch.bfloat16).to(device).eval()
    return te, te2, CLIPTokenizer.from_pretrained(flux_dir, subfolder="tokenizer"), T5TokenizerFast.from_pretrained(flux_dir, subfolder="tokenizer_2")


class TextCache:
    """prompt -> (T5 embeds [Lt, J] bf16, pooled [P] bf16).  Order: memory; ``<model_root>/prompt_cache/<sha1>.pt``;
    the caller's text encoders if given (``encoders`` = (text_encoder, text_encoder_2, tokenizer, tokenizer_2), result
    written back to the cache file); synthetic mode derives seeded stand-ins from the prompt hash."""

    def __init__(self, model_root: str, synthetic: bool, Lt: int, J: int, P: int, device, encoders=None, loader=None):
        self.root, self.synthetic, self.Lt, self.J, self.P, self.dev = model_root, synthetic, Lt, J, P, device
        self.encoders = encoders if encoders is not None and all(e is not None for e in encoders) else None
        self.loader = loader            # () -> encoders tuple, called on the first cache miss only (T5-XXL is 9.5 GB)
        self._mem: dict = {}

    def get(self, prompt: str, prompt_2: str = ""):
        key = hashlib.sha1(f"{prompt}\x00{prompt_2}".encode()).hexdigest()
        if key not in self._mem:
            path = os.path.join(self.root, "prompt_cache", key + ".pt")
            if os.path.exists(path):
                d = torch.load(path, map_location="cpu")
                t5, pooled = d["prompt_embeds"].reshape(-1, self.J), d["pooled_prompt_embeds"].reshape(-1)
            elif self.encoders is not None or (self.loader is not None and not self.synthetic):
                if self.encoders is None:
                    self.encoders = self.loader()
                t5, pooled = encode_prompt_with(*self.encoders, prompt, prompt_2, self.Lt)
                try:
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    torch.save({"prompt": prompt, "prompt_2": prompt_2, "prompt_embeds": t5.to(torch.bfloat16),
                                "pooled_prompt_embeds": pooled.to(torch.bfloat16)}, path)
                except OSError as e:        # read-only model directory: keep the encoding in memory only
                    print(f"prompt cache not written ({e})")
            elif self.synthetic:
                g = torch.Generator().manual_seed(int(key[:8], 16))
                t5, pooled = torch.randn(self.Lt, self.J, generator=g), torch.randn(self.P, generator=g)
            else:
                raise FileNotFoundError(f"no cached text encoding for prompt {prompt!r}: expected {path} "
                                        "(dict with prompt_embeds [512,4096], pooled_prompt_embeds [768]), or pass the text "
                                        "encoders / tokenizers to FluxPriorReduxPipeline.from_pretrained as the reference does")
            self._mem[key] = (t5.to(self.dev, torch.bfloat16).contiguous(), pooled.to(self.dev, torch.bfloat16).contiguous())
        return self._mem[key]


class FluxTxt2ImgHIP:
    """``FluxPipeline.__call__(prompt_embeds=…, pooled_prompt_embeds=…, guidance_scale, num_inference_steps, height,
    width, generator)`` of stage 2 (batch_generate_flux_kshot.py:467-474)."""

    def __init__(self, transformer: FluxTransformerHIP, vae: "vae_mod.FluxVaeHIP", use_graph: bool = True):
        self.tr, self.vae, self.dev, self.use_graph = transformer, vae, transformer.device, use_graph

    def __call__(self, prompt_embeds, pooled, *, height: int, width: int, guidance_scale: float, num_inference_steps: int,
                 noise_tokens: torch.Tensor, on_step=None) -> torch.Tensor:
        """``on_step(i, latents)``: called after every Euler update with a bf16 [B, n_tok, 64] COPY of the packed latents
        (what diffusers hands ``callback_on_step_end`` as ``latents``); used by scripts/accept_real_weights.py"""
        B = prompt_embeds.shape[0]
        h, w = height // 16, width // 16
        key = (B, h, w, prompt_embeds.shape[1])
        if getattr(self, "_key", None) != key:      # stable addresses: one captured graph serves every call of this shape
            bf = dict(dtype=torch.bfloat16, device=self.dev)
            self._lat, self._pe, self._pp = torch.empty((B, h * w, 64), **bf), torch.empty_like(prompt_embeds, **bf), torch.empty_like(pooled, **bf)
            self._key = key
        self._lat.copy_(noise_tokens); self._pe.copy_(prompt_embeds); self._pp.copy_(pooled)
        lat, prompt_embeds, pooled = self._lat, self._pe, self._pp
        sigmas, timesteps = flow_sigmas(num_inference_steps, h * w)
        img_ids, txt_ids = latent_image_ids(h, w), torch.zeros(prompt_embeds.shape[1], 3)
        guidance = torch.full((B,), float(guidance_scale)) if self.tr.cfg.guidance_embeds else None
        for i in range(num_inference_steps):
            t = torch.full((B,), model_timestep(timesteps[i]))
            fwd = self.tr.forward_graphed if self.use_graph else self.tr.forward
            v = fwd(lat, prompt_embeds, pooled, t, img_ids, txt_ids, guidance)
            ops.flow_euler_rows(lat, v, B * h * w, 64, 64, 64, float(sigmas[i + 1] - sigmas[i]))
            if on_step is not None:
                on_step(i, lat.clone())
        return self.vae.decode_tokens(lat, B, h, w, ld=64).clone()      # the decoder's buffer is reused by the next call


class Engine:
    """Everything stage 2 / stage 3 need, loaded once."""

    def __init__(self, kind: str, model_root: str = "./model", synthetic: bool = False, tiny: bool = False, device="cuda",
                 seed: int = 0):
        assert kind in ("dev", "fill")
        dev = torch.device(device)
        self.dev, self.kind = dev, kind
        fkw = dict(TINY["flux"]) if tiny else {}
        cfg = FluxConfig(in_channels=384 if kind == "fill" else 64, **fkw)
        vcfg = vae_mod.VaeConfig(**(TINY["vae"] if tiny else {}))
        vitcfg = vit_mod.VitConfig(**TINY["vit"]) if tiny else vit_mod.VitConfig.siglip_so400m()
        flux_dir = os.path.join(model_root, "FLUX.1-Fill-dev" if kind == "fill" else "FLUX.1-dev")
        redux_dir = os.path.join(model_root, "FLUX.1-Redux-dev")
        if synthetic:
            tp = init_params(cfg, seed=seed, device=dev)
            vp = vae_mod.init_params(vcfg, seed=seed + 1, device=dev)
            vitp = vit_mod.init_generic_params(vitcfg, seed + 2, device=dev)
            rp = redux_mod.init_redux_params(vitcfg.hidden, cfg.joint_attention_dim, seed=seed + 3, device=dev)
        else:
            for d in (flux_dir, redux_dir):
                if not os.path.isdir(d):
                    raise FileNotFoundError(f"{d} not found (pass --synthetic-weights to run with seeded random weights)")
            cfg = FluxConfig.from_json(os.path.join(flux_dir, "transformer", "config.json"))
            tp = load_safetensors_dir(os.path.join(flux_dir, "transformer"))
            vp = load_safetensors_dir(os.path.join(flux_dir, "vae"))
            vitp = vit_mod.siglip_to_generic(load_safetensors_dir(os.path.join(redux_dir, "image_encoder")), vitcfg)
            rp = load_safetensors_dir(os.path.join(redux_dir, "image_embedder"))
        tr = FluxTransformerHIP(cfg, tp, dev)
        vae = vae_mod.FluxVaeHIP(vcfg, vp, dev)
        del tp, vp
        self.cfg, self.vit_cfg = cfg, vitcfg
        self.prior = redux_mod.ReduxPriorHIP(vitcfg, vitp, rp, dev)
        self.pipe = FluxFillHIP(tr, vae) if kind == "fill" else FluxTxt2ImgHIP(tr, vae)
        self.text = TextCache(model_root, synthetic, TINY["t5_tokens"] if tiny else redux_mod.T5_TOKENS, cfg.joint_attention_dim,
                              cfg.pooled_projection_dim, dev,
                              loader=(lambda: load_text_encoders(flux_dir, dev)) if os.path.isdir(os.path.join(flux_dir, "text_encoder_2")) else None)

    def prior_embeds(self, pil_images, prompt: str, embeds_scale, pooled_scale):
        """pipe_prior_redux(images, prompt=…, prompt_2="", prompt_embeds_scale=…, pooled_prompt_embeds_scale=…)"""
        t5, pooled = self.text.get(prompt, "")
        imgs = siglip_input_device(pil_images, self.vit_cfg.image_size, self.dev)
        return self.prior(imgs, t5, pooled, embeds_scale, pooled_scale, group=len(pil_images))
