"""oracle/vit.py — ViT oracles built on the UPSTREAM `transformers` modelling code (importable here,
v5.x vs the reference's pinned 4.46.3: same architectures) loaded with the build's generic
parameter dict.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

SigLIP: ``SiglipVisionModel(...).last_hidden_state`` as FluxPriorReduxPipeline.encode_image uses it
(inside pipe_prior_redux, outpainting_updown_sampling_redux.py:1237-1243).
CLIP: ``CLIPVisionModelWithProjection(...).image_embeds`` == openai ``model.encode_image``
(retrieval/clip100_resnet_style_all_shots.py:171).
"""
from __future__ import annotations

import torch


def _layers_from_generic(sd, g, prefix, n_layers, names):
    for i in range(n_layers):
        for a, b in names:
            sd[f"{prefix}{i}.{b}.weight"] = g[f"l{i}.{a}.weight"]
            sd[f"{prefix}{i}.{b}.bias"] = g[f"l{i}.{a}.bias"]


def _load(m, sd, allow_missing):
    """load ``sd`` into ``m`` tolerating the ``vision_model.`` prefix difference between transformers majors"""
    have = set(m.state_dict().keys())
    fixed = {}
    for k, v in sd.items():
        if k in have:
            fixed[k] = v
        elif k.startswith("vision_model.") and k[len("vision_model."):] in have:
            fixed[k[len("vision_model."):]] = v
        else:
            fixed[k] = v
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in fixed.items()}, strict=False)
    assert not unexpected and all(any(a in k for a in allow_missing) for k in missing), (missing, unexpected)


def siglip_last_hidden_state(g: dict, image_size, patch, hidden, heads, layers, intermediate, pixel_values, dtype):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    cfg = SiglipVisionConfig(hidden_size=hidden, intermediate_size=intermediate, num_hidden_layers=layers,
                             num_attention_heads=heads, image_size=image_size, patch_size=patch,
                             hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    cfg.vision_use_head = False
    m = SiglipVisionModel(cfg).eval()
    sd = {"vision_model.embeddings.patch_embedding.weight": g["patch.weight"].view(hidden, 3, patch, patch),
          "vision_model.embeddings.patch_embedding.bias": g["patch.bias"],
          "vision_model.embeddings.position_embedding.weight": g["pos"],
          "vision_model.post_layernorm.weight": g["ln_post.weight"], "vision_model.post_layernorm.bias": g["ln_post.bias"]}
    _layers_from_generic(sd, g, "vision_model.encoder.layers.", layers,
                         (("ln1", "layer_norm1"), ("ln2", "layer_norm2"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                          ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")))
    _load(m, sd, ("head",))
    m = m.to(dtype)
    with torch.no_grad():
        return m(pixel_values=pixel_values.to(dtype)).last_hidden_state


def clip_image_embeds(g: dict, image_size, patch, hidden, heads, layers, intermediate, proj_dim, pixel_values, dtype):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=intermediate, num_hidden_layers=layers,
                           num_attention_heads=heads, image_size=image_size, patch_size=patch, hidden_act="quick_gelu",
                           layer_norm_eps=1e-5, projection_dim=proj_dim)
    m = CLIPVisionModelWithProjection(cfg).eval()
    v = "vision_model."
    sd = {v + "embeddings.patch_embedding.weight": g["patch.weight"].view(hidden, 3, patch, patch),
          v + "embeddings.class_embedding": g["cls"], v + "embeddings.position_embedding.weight": g["pos"],
          v + "pre_layrnorm.weight": g["ln_pre.weight"], v + "pre_layrnorm.bias": g["ln_pre.bias"],
          v + "post_layernorm.weight": g["ln_post.weight"], v + "post_layernorm.bias": g["ln_post.bias"],
          "visual_projection.weight": g["proj"].t()}
    _layers_from_generic(sd, g, v + "encoder.layers.", layers,
                         (("ln1", "layer_norm1"), ("ln2", "layer_norm2"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                          ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")))
    _load(m, sd, ("position_ids",))
    m = m.to(dtype)
    with torch.no_grad():
        return m(pixel_values=pixel_values.to(dtype)).image_embeds


def normalize_u8(img_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """uint8 [B,H,W,3] -> float32 NCHW ((u8/255 - mean)/std): SiglipImageProcessor / clip preprocess after resize"""
    x = img_u8.float() / 255.0
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    return x.permute(0, 3, 1, 2).contiguous()
