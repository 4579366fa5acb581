"""The real-checkpoint path (no checkpoints exist offline): tiny random weights are written in the directory layout and
key naming of the FLUX.1 / Redux / openai-CLIP checkpoints, loaded through the product loaders, and must give the same
bits as the host classes fed with the same tensors directly."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _siglip_hf_names(generic: dict, layers: int) -> dict:
    p = "vision_model."
    P = int(round((generic["patch.weight"].shape[1] // 3) ** 0.5))
    sd = {p + "embeddings.patch_embedding.weight": generic["patch.weight"].view(-1, 3, P, P).contiguous(),
          p + "embeddings.patch_embedding.bias": generic["patch.bias"], p + "embeddings.position_embedding.weight": generic["pos"],
          p + "post_layernorm.weight": generic["ln_post.weight"], p + "post_layernorm.bias": generic["ln_post.bias"],
          p + "head.probe": torch.zeros(1, 1, generic["pos"].shape[1], dtype=torch.bfloat16)}        # unused pooling head
    for i in range(layers):
        for a, b in (("ln1", "layer_norm1"), ("ln2", "layer_norm2"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                     ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            for t in ("weight", "bias"):
                sd[f"{p}encoder.layers.{i}.{b}.{t}"] = generic[f"l{i}.{a}.{t}"]
    return sd


def test_engine_loads_checkpoint_directories(gpu, tmp_path):
    from safetensors.torch import save_file
    from PIL import Image
    from domain_rag_amd import redux, vae, vit
    from domain_rag_amd.engine import TINY, Engine, generator_noise, pack_noise
    from domain_rag_amd.fill_pipeline import FluxFillHIP
    from domain_rag_amd.flux import FluxTransformerHIP
    from domain_rag_amd.flux_params import FluxConfig, init_params
    cfg = FluxConfig(in_channels=384, **TINY["flux"])
    tp = init_params(cfg, seed=0)
    vcfg = vae.VaeConfig(**TINY["vae"])
    vp = vae.init_params(vcfg, seed=1)
    vitcfg = vit.VitConfig(**TINY["vit"])
    vitp = vit.init_generic_params(vitcfg, 2)
    rp = redux.init_redux_params(vitcfg.hidden, cfg.joint_attention_dim, seed=3)
    root = tmp_path / "model"
    fill, rdx = root / "FLUX.1-Fill-dev", root / "FLUX.1-Redux-dev"
    for d in (fill / "transformer", fill / "vae", rdx / "image_encoder", rdx / "image_embedder"):
        d.mkdir(parents=True)
    keys = sorted(tp)
    half = len(keys) // 2                                        # two shards, like diffusion_pytorch_model-0000X-of-0000Y
    save_file({k: tp[k].contiguous() for k in keys[:half]}, str(fill / "transformer" / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: tp[k].contiguous() for k in keys[half:]}, str(fill / "transformer" / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    json.dump(dict(in_channels=384, out_channels=None, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                   num_attention_heads=cfg.num_attention_heads, attention_head_dim=128, joint_attention_dim=cfg.joint_attention_dim,
                   pooled_projection_dim=cfg.pooled_projection_dim, guidance_embeds=True, axes_dims_rope=[16, 56, 56]),
              open(fill / "transformer" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in vp.items()}, str(fill / "vae" / "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in _siglip_hf_names(vitp, vitcfg.layers).items()}, str(rdx / "image_encoder" / "model.safetensors"))
    save_file({k: v.contiguous() for k, v in rp.items()}, str(rdx / "image_embedder" / "diffusion_pytorch_model.safetensors"))
    # prompt cache entry for "" (no text encoders in this layout)
    import hashlib
    (root / "prompt_cache").mkdir()
    g = torch.Generator().manual_seed(9)
    t5, pooled = torch.randn(TINY["t5_tokens"], cfg.joint_attention_dim, generator=g).bfloat16(), torch.randn(cfg.pooled_projection_dim, generator=g).bfloat16()
    torch.save({"prompt_embeds": t5, "pooled_prompt_embeds": pooled}, root / "prompt_cache" / (hashlib.sha1(b"\x00").hexdigest() + ".pt"))

    eng = Engine("fill", str(root), synthetic=False, tiny=True, device=gpu)
    assert eng.cfg.in_channels == 384 and eng.cfg.num_layers == cfg.num_layers
    bg = Image.fromarray(np.random.default_rng(0).integers(0, 256, (70, 90, 3), dtype=np.uint8))
    pe, pp = eng.prior_embeds([bg], "", [1.2], [1.0])
    H, W = 64, 96
    img = torch.randint(0, 256, (1, H, W, 3), generator=g, dtype=torch.uint8).to(gpu)
    msk = torch.full((1, H, W), 255, dtype=torch.uint8); msk[:, 10:30, 20:50] = 0
    en, nz, mn = generator_noise(5, 1, H, W, 3)
    kw = dict(guidance_scale=30.0, num_inference_steps=2, strength=1.0, enc_noise=en.to(gpu), masked_enc_noise=mn.to(gpu), noise_tokens=pack_noise(nz).to(gpu))
    out = eng.pipe(img, msk.to(gpu), pe, pp, **kw)
    # the same tensors handed to the host classes directly
    from domain_rag_amd.engine import siglip_input_device
    prior = redux.ReduxPriorHIP(vitcfg, vitp, rp, gpu)
    pe2, pp2 = prior(siglip_input_device([bg], vitcfg.image_size, gpu), t5.to(gpu), pooled.to(gpu), [1.2], [1.0], group=1)
    assert torch.equal(pe, pe2) and torch.equal(pp, pp2)
    direct = FluxFillHIP(FluxTransformerHIP(cfg, tp, gpu), vae.FluxVaeHIP(vcfg, vp, gpu))
    assert torch.equal(out, direct(img, msk.to(gpu), pe2, pp2, **kw))


def _tiny_openai_clip(seed=0, D=128, layers=2, P=32, grid=2, proj=64):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=0.05: (torch.randn(*s, generator=g) * sc).half()
    sd = {"visual.conv1.weight": rn(D, 3, P, P), "visual.class_embedding": rn(D), "visual.positional_embedding": rn(grid * grid + 1, D),
          "visual.ln_pre.weight": 1 + rn(D), "visual.ln_pre.bias": rn(D), "visual.ln_post.weight": 1 + rn(D), "visual.ln_post.bias": rn(D),
          "visual.proj": rn(D, proj), "logit_scale": torch.tensor(4.6), "text_projection": rn(8, 8)}
    for i in range(layers):
        s = f"visual.transformer.resblocks.{i}."
        sd.update({s + "attn.in_proj_weight": rn(3 * D, D), s + "attn.in_proj_bias": rn(3 * D), s + "attn.out_proj.weight": rn(D, D),
                   s + "attn.out_proj.bias": rn(D), s + "ln_1.weight": 1 + rn(D), s + "ln_1.bias": rn(D), s + "ln_2.weight": 1 + rn(D),
                   s + "ln_2.bias": rn(D), s + "mlp.c_fc.weight": rn(4 * D, D), s + "mlp.c_fc.bias": rn(4 * D),
                   s + "mlp.c_proj.weight": rn(D, 4 * D), s + "mlp.c_proj.bias": rn(D)})
    return sd


def test_load_clip_from_state_dict_file_and_jit_archive(gpu, tmp_path):
    """openai distributes ViT-B-32.pt as a TorchScript archive; a plain torch.save'd state_dict is accepted too; the ViT
    dimensions are read off the tensors like clip.model.build_model does"""
    from domain_rag_amd import retrieval as R
    sd = _tiny_openai_clip()
    torch.save(sd, tmp_path / "sd.pt")

    class Holder(torch.nn.Module):          # a scriptable module whose state_dict has the openai key names
        def __init__(self):
            super().__init__()
            for k, v in sd.items():
                mod = self
                *path, leaf = k.split(".")
                for part in path:
                    if not hasattr(mod, part):
                        setattr(mod, part, torch.nn.Module())
                    mod = getattr(mod, part)
                mod.register_buffer(leaf, v.clone())

        def forward(self, x):
            return x
    torch.jit.script(Holder()).save(str(tmp_path / "ViT-B-32.pt"))
    x = torch.randint(0, 256, (3, 64, 64, 3), generator=torch.Generator().manual_seed(1), dtype=torch.uint8).to(gpu)
    outs = []
    for w in (sd, str(tmp_path / "sd.pt"), str(tmp_path / "ViT-B-32.pt")):
        model, _ = R.load_clip("ViT-B/32", device=gpu, weights=w)
        assert model.visual.cfg.hidden == 128 and model.visual.cfg.layers == 2 and model.visual.cfg.image_size == 64 and model.visual.cfg.proj_dim == 64
        outs.append(model.encode_image(x).clone())
    assert outs[0].shape == (3, 64) and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # and against the transformers CLIP vision tower with the same weights (the ViT oracle)
    from oracle import vit as ovit
    from domain_rag_amd.vit import VitConfig, openai_clip_to_generic
    cfg = VitConfig.from_openai_state_dict(sd)
    ref = ovit.clip_image_embeds(openai_clip_to_generic(sd, cfg), cfg.image_size, cfg.patch_size, cfg.hidden, cfg.heads, cfg.layers,
                                   cfg.intermediate, cfg.proj_dim, ovit.normalize_u8(x.cpu(), cfg.mean, cfg.std), torch.float32)
    err = (outs[0].cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-2, err
