"""ViT image encoders on the HIP path: SigLIP-so400m/14-384 (the Redux image encoder) and CLIP
ViT-B/32 (retrieval).  One generic pre-LN encoder; every matmul is the MFMA GEMM, attention is the
fused MFMA kernel with heads zero-padded to 128 columns (72 -> 128 for SigLIP, 64 -> 128 for CLIP:
padded q/k columns contribute 0 to the scores, padded v columns are dropped by zero rows of the
out-projection), LayerNorm is the one-wave-per-row kernel.

Replaces ``SiglipVisionModel`` (transformers 4.46.3, inside ``pipe_prior_redux(...)``:
batch_generate_flux_kshot.py:459-465, outpainting_updown_sampling_redux.py:1237-1243) and
``clip.model.VisionTransformer`` (openai/CLIP@dcba3cb, ``model.encode_image``:
retrieval/clip100_resnet_style_all_shots.py:171,284,337,948).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import ops


@dataclass
class VitConfig:
    image_size: int = 384
    patch_size: int = 14
    hidden: int = 1152
    heads: int = 16
    layers: int = 27
    intermediate: int = 4304
    act: int = ops.ACT_GELU_TANH
    ln_eps: float = 1e-6
    cls_token: bool = False          # CLIP: class embedding + ln_pre + ln_post(cls) + projection
    patch_bias: bool = True
    proj_dim: int = 0
    mean: tuple = (0.5, 0.5, 0.5)
    std: tuple = (0.5, 0.5, 0.5)

    @classmethod
    def siglip_so400m(cls) -> "VitConfig":
        return cls()

    @classmethod
    def clip_vit_b32(cls) -> "VitConfig":
        return cls(image_size=224, patch_size=32, hidden=768, heads=12, layers=12, intermediate=3072,
                   act=ops.ACT_QUICK_GELU, ln_eps=1e-5, cls_token=True, patch_bias=False, proj_dim=512,
                   mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711))

    @classmethod
    def from_openai_state_dict(cls, sd: dict) -> "VitConfig":
        """ViT dimensions of an openai/CLIP state_dict (how ``clip.model.build_model`` infers them)"""
        D, _, P, _ = sd["visual.conv1.weight"].shape
        layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
        grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        c = cls.clip_vit_b32()
        return cls(image_size=grid * P, patch_size=P, hidden=D, heads=D // 64, layers=layers,
                   intermediate=sd["visual.transformer.resblocks.0.mlp.c_fc.weight"].shape[0], act=c.act, ln_eps=c.ln_eps, cls_token=True,
                   patch_bias=False, proj_dim=sd["visual.proj"].shape[1], mean=c.mean, std=c.std)

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + (1 if self.cls_token else 0)


def _pad(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _pad_heads_rows(w: torch.Tensor, heads: int) -> torch.Tensor:
    """[heads*hd, D] (or [heads*hd]) -> [heads*128, D] with zero rows for the padded head columns"""
    hd = w.shape[0] // heads
    out = torch.zeros((heads, 128) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:, :hd] = w.view((heads, hd) + tuple(w.shape[1:]))
    return out.view((heads * 128,) + tuple(w.shape[1:]))


def siglip_to_generic(sd: dict, cfg: VitConfig) -> dict:
    """transformers SiglipVisionModel state_dict -> generic names used by VitHIP"""
    p = "vision_model."
    g = {"patch.weight": sd[p + "embeddings.patch_embedding.weight"].flatten(1),
         "patch.bias": sd[p + "embeddings.patch_embedding.bias"],
         "pos": sd[p + "embeddings.position_embedding.weight"],
         "ln_post.weight": sd[p + "post_layernorm.weight"], "ln_post.bias": sd[p + "post_layernorm.bias"]}
    for i in range(cfg.layers):
        s = f"{p}encoder.layers.{i}."
        for a, b in (("ln1", "layer_norm1"), ("ln2", "layer_norm2"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                     ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            g[f"l{i}.{a}.weight"] = sd[s + b + ".weight"]
            g[f"l{i}.{a}.bias"] = sd[s + b + ".bias"]
    return g


def openai_clip_to_generic(sd: dict, cfg: VitConfig) -> dict:
    """openai/CLIP ``visual.*`` state_dict -> generic names"""
    D = cfg.hidden
    g = {"patch.weight": sd["visual.conv1.weight"].flatten(1), "cls": sd["visual.class_embedding"],
         "pos": sd["visual.positional_embedding"],
         "ln_pre.weight": sd["visual.ln_pre.weight"], "ln_pre.bias": sd["visual.ln_pre.bias"],
         "ln_post.weight": sd["visual.ln_post.weight"], "ln_post.bias": sd["visual.ln_post.bias"],
         "proj": sd["visual.proj"]}
    for i in range(cfg.layers):
        s = f"visual.transformer.resblocks.{i}."
        w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
        for j, n in enumerate("qkv"):
            g[f"l{i}.{n}.weight"], g[f"l{i}.{n}.bias"] = w[j * D:(j + 1) * D], b[j * D:(j + 1) * D]
        for a, bn in (("ln1", "ln_1"), ("ln2", "ln_2"), ("o", "attn.out_proj"), ("fc1", "mlp.c_fc"), ("fc2", "mlp.c_proj")):
            g[f"l{i}.{a}.weight"], g[f"l{i}.{a}.bias"] = sd[s + bn + ".weight"], sd[s + bn + ".bias"]
    return g


def init_generic_params(cfg: VitConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16) -> dict:
    """seeded synthetic weights in generic naming (no checkpoints offline)"""
    gen = torch.Generator(device=device).manual_seed(seed)
    D, F = cfg.hidden, cfg.intermediate
    kk = 3 * cfg.patch_size ** 2

    def rn(*shape, s=0.02):
        return (s * torch.randn(shape, generator=gen, device=device)).to(dtype)

    g = {"patch.weight": rn(D, kk, s=1.0 / math.sqrt(kk)), "pos": rn(cfg.tokens, D),
         "ln_post.weight": (1 + rn(D, s=0.1).float()).to(dtype), "ln_post.bias": rn(D)}
    if cfg.patch_bias:
        g["patch.bias"] = rn(D)
    if cfg.cls_token:
        g["cls"] = rn(D)
        g["ln_pre.weight"], g["ln_pre.bias"] = (1 + rn(D, s=0.1).float()).to(dtype), rn(D)
        g["proj"] = rn(D, cfg.proj_dim, s=1.0 / math.sqrt(D))
    for i in range(cfg.layers):
        for n in ("ln1", "ln2"):
            g[f"l{i}.{n}.weight"], g[f"l{i}.{n}.bias"] = (1 + rn(D, s=0.1).float()).to(dtype), rn(D)
        for n in ("q", "k", "v", "o"):
            g[f"l{i}.{n}.weight"], g[f"l{i}.{n}.bias"] = rn(D, D, s=1.0 / math.sqrt(D)), rn(D)
        g[f"l{i}.fc1.weight"], g[f"l{i}.fc1.bias"] = rn(F, D, s=1.0 / math.sqrt(D)), rn(F)
        g[f"l{i}.fc2.weight"], g[f"l{i}.fc2.bias"] = rn(D, F, s=1.0 / math.sqrt(F)), rn(D)
    return g


class VitHIP:
    def __init__(self, cfg: VitConfig, g: dict, device="cuda"):
        self.cfg, self.dev = cfg, torch.device(device)
        D, H, F = cfg.hidden, cfg.heads, cfg.intermediate
        if D % 64 or (D // H) > 128:
            raise ValueError("hidden must be a multiple of 64 and head_dim <= 128")
        self.hd = D // H
        Fp, Kp = _pad(F, 64), _pad(3 * cfg.patch_size ** 2, 64)
        self.Fp, self.Kp = Fp, Kp
        bf = dict(dtype=torch.bfloat16, device=self.dev)

        def dv(t):
            return t.to(**bf).contiguous()

        wp = torch.zeros((D, Kp), **bf)
        wp[:, : 3 * cfg.patch_size ** 2] = dv(g["patch.weight"])
        self.w_patch, self.b_patch = wp, (dv(g["patch.bias"]) if cfg.patch_bias else None)
        self.pos = dv(g["pos"])
        self.cls = dv(g["cls"]) if cfg.cls_token else None
        self.ln_pre = (dv(g["ln_pre.weight"]), dv(g["ln_pre.bias"])) if cfg.cls_token else None
        self.ln_post = (dv(g["ln_post.weight"]), dv(g["ln_post.bias"]))
        self.proj = dv(g["proj"]).t().contiguous() if cfg.proj_dim else None      # [proj_dim, D]
        self.layers = []
        for i in range(cfg.layers):
            q, k, v = (dv(g[f"l{i}.{n}.weight"]) for n in "qkv")
            bq, bk, bv = (dv(g[f"l{i}.{n}.bias"]) for n in "qkv")
            wqkv = torch.cat([_pad_heads_rows(q, H), _pad_heads_rows(k, H), _pad_heads_rows(v, H)], 0).contiguous()
            bqkv = torch.cat([_pad_heads_rows(bq, H), _pad_heads_rows(bk, H), _pad_heads_rows(bv, H)], 0).contiguous()
            wo = _pad_heads_rows(dv(g[f"l{i}.o.weight"]).t().contiguous(), H).t().contiguous()   # [D, H*128]
            w1 = torch.zeros((Fp, D), **bf); w1[:F] = dv(g[f"l{i}.fc1.weight"])
            b1 = torch.zeros((Fp,), **bf); b1[:F] = dv(g[f"l{i}.fc1.bias"])
            w2 = torch.zeros((D, Fp), **bf); w2[:, :F] = dv(g[f"l{i}.fc2.weight"])
            self.layers.append(dict(ln1=(dv(g[f"l{i}.ln1.weight"]), dv(g[f"l{i}.ln1.bias"])),
                                    ln2=(dv(g[f"l{i}.ln2.weight"]), dv(g[f"l{i}.ln2.bias"])),
                                    wqkv=wqkv, bqkv=bqkv, wo=wo, bo=dv(g[f"l{i}.o.bias"]),
                                    w1=w1, b1=b1, w2=w2, b2=dv(g[f"l{i}.fc2.bias"])))
        self._ws_B = None

    def _workspace(self, B):
        if self._ws_B == B:
            return self._ws
        cfg = self.cfg
        T, D, H = cfg.tokens, cfg.hidden, cfg.heads
        nP = T - (1 if cfg.cls_token else 0)
        bf = dict(dtype=torch.bfloat16, device=self.dev)
        tmpl = self.pos[None].expand(B, T, D).clone()      # clone: for B = 1 .contiguous() would alias self.pos and the += below would corrupt it
        if cfg.cls_token:
            tmpl[:, 0] += self.cls           # setup-time constant (class_embedding + positional_embedding[0])
        ws = dict(tmpl=tmpl, patches=torch.empty((B * nP, self.Kp), **bf), x=torch.empty((B, T, D), **bf),
                  nrm=torch.empty((B * T, D), **bf), qkv=torch.empty((B, T, 3 * H * 128), **bf),
                  vt=torch.empty((B, H, 128, _pad(T, 64)), **bf), att=torch.empty((B, T, H * 128), **bf),
                  hid=torch.empty((B * T, self.Fp), **bf), out=torch.empty((B * T, D), **bf))
        self._ws_B, self._ws = B, ws
        return ws

    def forward(self, img_u8: torch.Tensor) -> torch.Tensor:
        """uint8 RGB [B, S, S, 3] (already resized to cfg.image_size) -> SigLIP: bf16 [B, T, D]
        (last_hidden_state); CLIP: fp32 [B, proj_dim] (encode_image, not normalised)."""
        cfg = self.cfg
        B = img_u8.shape[0]
        T, D, H = cfg.tokens, cfg.hidden, cfg.heads
        c0 = 1 if cfg.cls_token else 0
        nP = T - c0
        ws = self._workspace(B)
        x, nrm, qkv, vt, att, hid = ws["x"], ws["nrm"], ws["qkv"], ws["vt"], ws["att"], ws["hid"]
        S = cfg.image_size
        if img_u8.dtype == torch.uint8:
            if tuple(img_u8.shape[1:]) != (S, S, 3):
                raise ValueError(f"expected uint8 [B,{S},{S},3]")
            ops.patchify(img_u8.contiguous(), ws["patches"], B, S, S, cfg.patch_size, self.Kp, cfg.mean, cfg.std)
        else:   # already normalised float NCHW (e.g. clip `preprocess` output)
            if tuple(img_u8.shape[1:]) != (3, S, S):
                raise ValueError(f"expected float [B,3,{S},{S}]")
            ops.patchify_f32(img_u8.float().contiguous(), ws["patches"], B, S, S, cfg.patch_size, self.Kp)
        if c0:
            x.copy_(ws["tmpl"])               # device memcpy: row 0 of every image = cls + pos[0]
        ops.gemm(ws["patches"], self.w_patch, out=x.view(-1)[c0 * D:], bias=self.b_patch, M=B * nP, lda=self.Kp,
                 c_rows_per_batch=nP, c_batch_stride=T * D, ldc=D, resid=ws["tmpl"].view(-1)[c0 * D:])
        M = B * T
        if self.ln_pre is not None:
            ops.layernorm(x, nrm, M, D, gamma=self.ln_pre[0], beta=self.ln_pre[1], eps=cfg.ln_eps)
            x.copy_(nrm.view(B, T, D))
        scale = 1.0 / math.sqrt(self.hd)
        HD = H * 128
        for L in self.layers:
            ops.layernorm(x, nrm, M, D, gamma=L["ln1"][0], beta=L["ln1"][1], eps=cfg.ln_eps)
            ops.gemm(nrm, L["wqkv"], out=qkv, bias=L["bqkv"], M=M, lda=D, ldc=3 * HD)
            ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, T, H, 3 * HD, 0)
            ops.attention(qkv, qkv.view(-1)[HD:], vt, att, B, T, H, 3 * HD, T * 3 * HD, HD, T * HD, scale)
            ops.gemm(att, L["wo"], out=x, bias=L["bo"], M=M, lda=HD, ldc=D, resid=x)
            ops.layernorm(x, nrm, M, D, gamma=L["ln2"][0], beta=L["ln2"][1], eps=cfg.ln_eps)
            ops.gemm(nrm, L["w1"], out=hid, bias=L["b1"], act=cfg.act, M=M, lda=D, ldc=self.Fp)
            ops.gemm(hid, L["w2"], out=x, bias=L["b2"], M=M, lda=self.Fp, ldc=D, resid=x)
        if cfg.cls_token:
            pooled = nrm.view(-1)[: B * D].view(B, D)
            ops.layernorm(x, pooled, B, D, gamma=self.ln_post[0], beta=self.ln_post[1], ldx=T * D, eps=cfg.ln_eps)
            return ops.gemm(pooled, self.proj, out_f32=True)
        ops.layernorm(x, ws["out"], M, D, gamma=self.ln_post[0], beta=self.ln_post[1], eps=cfg.ln_eps)
        return ws["out"].view(B, T, D)          # a view of the workspace: valid until the next call (the Redux prior consumes it at once)

    __call__ = forward


class ClipVitF32HIP:
    """CLIP's ``VisionTransformer`` in float32 on the GPU (csrc/vit_f32.hip + the f32-MFMA conv kernel): the arithmetic
    openai-CLIP itself uses on its CPU path (``clip.load`` calls ``model.float()`` there; on CUDA it keeps fp16 weights).
    Patch embedding = stride-P convolution over an NHWC frame with a zero 4th channel, every Linear a 1x1 convolution with
    bias / QuickGELU / residual in its epilogue, LayerNorm two-pass in fp32, attention per (image, head) in LDS.
    Takes the same generic parameter dict as VitHIP; class-token models with tokens <= 64 and head_dim <= 64 (ViT-B/32)."""

    def __init__(self, cfg: VitConfig, g: dict, device="cuda"):
        self.cfg, self.dev = cfg, torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("ClipVitF32HIP needs a GPU device (domain-rag_amd has no CPU path)")
        D, H, P = cfg.hidden, cfg.heads, cfg.patch_size
        if not cfg.cls_token or not cfg.proj_dim or cfg.tokens > 64 or D // H > 64 or D > 1024 or D % 4 or cfg.intermediate % 4:
            raise ValueError("ClipVitF32HIP covers CLIP-style towers with <= 64 tokens and head_dim <= 64 (ViT-B/32)")
        if cfg.act != ops.ACT_QUICK_GELU:
            raise ValueError("ClipVitF32HIP implements openai-CLIP's QuickGELU MLP")
        self.hd = D // H

        def dv(t):
            return t.detach().to(device=self.dev, dtype=torch.float32).contiguous()

        wp = torch.zeros((D, P, P, 4), dtype=torch.float32, device=self.dev)          # conv1.weight [D,3,P,P] -> [D,P,P,4]
        wp[..., :3] = dv(g["patch.weight"]).view(D, 3, P, P).permute(0, 2, 3, 1)
        # the P pixels x 4 channels of a patch row are contiguous in the NHWC frame: a P x 1 kernel over 4P "channels"
        self.w_patch = wp.view(D, P, 1, 4 * P)
        self.cls, self.pos = dv(g["cls"]), dv(g["pos"])
        self.ln_pre = (dv(g["ln_pre.weight"]), dv(g["ln_pre.bias"]))
        self.ln_post = (dv(g["ln_post.weight"]), dv(g["ln_post.bias"]))
        self.proj = dv(g["proj"]).t().contiguous()                                    # [proj_dim, D]
        self.layers = []
        for i in range(cfg.layers):
            self.layers.append(dict(
                ln1=(dv(g[f"l{i}.ln1.weight"]), dv(g[f"l{i}.ln1.bias"])), ln2=(dv(g[f"l{i}.ln2.weight"]), dv(g[f"l{i}.ln2.bias"])),
                wqkv=torch.cat([dv(g[f"l{i}.{n}.weight"]) for n in "qkv"], 0).contiguous(),
                bqkv=torch.cat([dv(g[f"l{i}.{n}.bias"]) for n in "qkv"], 0).contiguous(),
                wo=dv(g[f"l{i}.o.weight"]), bo=dv(g[f"l{i}.o.bias"]),
                w1=dv(g[f"l{i}.fc1.weight"]), b1=dv(g[f"l{i}.fc1.bias"]), w2=dv(g[f"l{i}.fc2.weight"]), b2=dv(g[f"l{i}.fc2.bias"])))
        self._ws_B, self._ws = None, None

    def _workspace(self, B):
        if self._ws_B != B:
            cfg = self.cfg
            T, D, S = cfg.tokens, cfg.hidden, cfg.image_size
            f = dict(dtype=torch.float32, device=self.dev)
            self._ws = dict(frame=torch.empty((B, S, S, 4), **f), emb=torch.empty((B * (T - 1), D), **f), x=torch.empty((B * T, D), **f),
                            nrm=torch.empty((B * T, D), **f), qkv=torch.empty((B * T, 3 * D), **f), att=torch.empty((B * T, D), **f),
                            hid=torch.empty((B * T, cfg.intermediate), **f), pooled=torch.empty((B, D), **f))
            self._ws_B = B
        return self._ws

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        """uint8 RGB [B,S,S,3] or normalised float [B,3,S,S] (``preprocess`` output) -> fp32 [B, proj_dim] (encode_image)"""
        cfg = self.cfg
        B, S, P = img.shape[0], cfg.image_size, cfg.patch_size
        T, D, H, F = cfg.tokens, cfg.hidden, cfg.heads, cfg.intermediate
        if img.dtype == torch.uint8:
            if tuple(img.shape[1:]) != (S, S, 3):
                raise ValueError(f"expected uint8 [B,{S},{S},3]")
        elif tuple(img.shape[1:]) != (3, S, S):
            raise ValueError(f"expected float [B,3,{S},{S}]")
        ws = self._workspace(B)
        x, nrm, qkv, att, hid = ws["x"], ws["nrm"], ws["qkv"], ws["att"], ws["hid"]
        ops.vit_prepare(img.contiguous() if img.dtype == torch.uint8 else img.float().contiguous(), ws["frame"], cfg.mean, cfg.std)
        gp = S // P
        ops.conv2d_f32(ws["frame"], self.w_patch, ws["emb"], B=B, Hi=S, Wi=S, Ho=gp, Wo=gp, Cin=4 * P, ldx=4, ldy=D, stride=P)
        ops.clip_embed_ln(ws["emb"], self.cls, self.pos, self.ln_pre[0], self.ln_pre[1], x, B, T, D, cfg.ln_eps)
        M = B * T
        scale = 1.0 / math.sqrt(self.hd)
        for L in self.layers:
            ops.layernorm_f32(x, nrm, L["ln1"][0], L["ln1"][1], M, D, cfg.ln_eps)
            ops.linear_f32(nrm, L["wqkv"], qkv, M, ldx=D, ldy=3 * D, bias=L["bqkv"])
            ops.attention_small_f32(qkv, att, B, T, H, self.hd, 3 * D, D, scale)
            ops.linear_f32(att, L["wo"], x, M, ldx=D, ldy=D, bias=L["bo"], resid=x, ld_res=D)
            ops.layernorm_f32(x, nrm, L["ln2"][0], L["ln2"][1], M, D, cfg.ln_eps)
            ops.linear_f32(nrm, L["w1"], hid, M, ldx=D, ldy=F, bias=L["b1"], act=ops.CONV_ACT_QUICK_GELU)
            ops.linear_f32(hid, L["w2"], x, M, ldx=F, ldy=D, bias=L["b2"], resid=x, ld_res=D)
        ops.layernorm_f32(x, ws["pooled"], self.ln_post[0], self.ln_post[1], B, D, cfg.ln_eps, ldx=T * D, ldy=D)
        out = torch.empty((B, cfg.proj_dim), dtype=torch.float32, device=self.dev)
        return ops.linear_f32(ws["pooled"], self.proj, out, B, ldx=D, ldy=cfg.proj_dim)

    __call__ = forward
