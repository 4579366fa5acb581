"""FluxTransformerHIP — host driver of the Flux DiT forward on MI355X.

Mirrors ``FluxTransformer2DModel.forward`` (diffusers 0.33.1; the call the reference reaches on
every denoise step from batch_generate_flux_kshot.py:467-474 and
outpainting_updown_sampling_redux.py:1246-1257).  All arithmetic is in libdomainrag_hip.so; torch
only owns the device buffers.  MI355X-first layout decisions:

* text and image streams live in ONE joint activation buffer ``x[B, S_txt+S_img, D]`` for the
  whole forward (the GEMM / LayerNorm kernels address "batched rows"), so there are no concat /
  split copies between double-stream and single-stream blocks;
* q/k/v projections are fused into one [3D, D] weight; single-stream blocks write attention and
  MLP outputs side by side into one [M, 5D] buffer that feeds proj_out directly;
* every block's AdaLN modulation is computed by ONE GEMM per forward (all modulation weights are
  stacked), instead of 76 tiny launches;
* workspaces are allocated once per (B, S) and reused: 288 GB of HBM keeps weights (23.7 GB bf16)
  plus B=8 activations resident.
"""
from __future__ import annotations

import math

import torch

import os

from . import ops
from .flux_params import FluxConfig

_SEPARATE_QPREP = os.environ.get("DRAG_QPREP_SEPARATE", "") not in ("", "0")     # measurement switches, read once
# single blocks: to_q|k|v + proj_mlp as ONE two-destination launch.  At the headline's 42 696 rows it saves a partial tile round
# (54.8 instead of 24 + 32) and measured nothing (same-box A/B, two rounds: fused 0.4440 / 0.4435 vs separate 0.4468 / 0.4454 images/s);
# at BASELINE configs[1]'s 1536 rows the fused launch is two exact rounds of 256x256 tiles (1400 TFLOP/s) where the separate ones are
# one round + 4.5 rounds of small tiles (1132 / 800-1000).  So the choice follows the library's own cost model per launch shape
# (ops.gemm_cost: fused when it is >= 10 % cheaper); DRAG_QKV_MLP_FUSED=1 / =0 force it on / off.
_FUSED_QKV_MLP = {"": None, "0": False}.get(os.environ.get("DRAG_QKV_MLP_FUSED", ""), True)


def rope_tables(ids: torch.Tensor, axes_dims=(16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed on the host in float64 (as diffusers does off-MPS) -> fp32 [S, 64] cos / sin."""
    cos, sin = [], []
    pos = ids.detach().to("cpu", torch.float64)
    for i, d in enumerate(axes_dims):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i], freqs)
        cos.append(ang.cos().float())
        sin.append(ang.sin().float())
    return torch.cat(cos, dim=1).contiguous(), torch.cat(sin, dim=1).contiguous()


def latent_image_ids(h: int, w: int) -> torch.Tensor:
    ids = torch.zeros(h, w, 3)
    ids[..., 1] = torch.arange(h)[:, None]
    ids[..., 2] = torch.arange(w)[None, :]
    return ids.reshape(h * w, 3)


class FluxTransformerHIP:
    def __init__(self, cfg: FluxConfig, params: dict, device="cuda"):
        if cfg.attention_head_dim != 128:
            raise ValueError("the HIP attention kernel is specialised for head_dim 128 (all FLUX.1 models)")
        self.cfg = cfg
        self.device = torch.device(device)
        D = cfg.dim
        dev = self.device

        def g(name):
            return params[name].to(dev, torch.bfloat16).contiguous()

        def cat(names):
            return torch.cat([params[n].to(dev, torch.bfloat16) for n in names], dim=0).contiguous()

        self.w = {}
        for n in ("x_embedder", "context_embedder", "proj_out",
                  "time_text_embed.timestep_embedder.linear_1", "time_text_embed.timestep_embedder.linear_2",
                  "time_text_embed.text_embedder.linear_1", "time_text_embed.text_embedder.linear_2"):
            self.w[n + ".weight"], self.w[n + ".bias"] = g(n + ".weight"), g(n + ".bias")
        if cfg.guidance_embeds:
            for n in ("time_text_embed.guidance_embedder.linear_1", "time_text_embed.guidance_embedder.linear_2"):
                self.w[n + ".weight"], self.w[n + ".bias"] = g(n + ".weight"), g(n + ".bias")

        # ---- stacked modulation weights: [double i: norm1 (6D) | norm1_context (6D)]*L, [single: 3D]*Ls, norm_out 2D
        mod_w, mod_b = [], []
        self.mod_off = {}
        off = 0
        for i in range(cfg.num_layers):
            for nm in ("norm1", "norm1_context"):
                key = f"transformer_blocks.{i}.{nm}.linear"
                mod_w.append(key + ".weight"); mod_b.append(key + ".bias")
                self.mod_off[(i, nm)] = off
                off += 6 * D
        for i in range(cfg.num_single_layers):
            key = f"single_transformer_blocks.{i}.norm.linear"
            mod_w.append(key + ".weight"); mod_b.append(key + ".bias")
            self.mod_off[("s", i)] = off
            off += 3 * D
        mod_w.append("norm_out.linear.weight"); mod_b.append("norm_out.linear.bias")
        self.mod_off["out"] = off
        off += 2 * D
        self.mod_total = off
        self.mod_w, self.mod_b = cat(mod_w), cat(mod_b)

        self.double = []
        for i in range(cfg.num_layers):
            p = f"transformer_blocks.{i}."
            self.double.append(dict(
                wqkv=cat([p + "attn.to_q.weight", p + "attn.to_k.weight", p + "attn.to_v.weight"]),
                bqkv=cat([p + "attn.to_q.bias", p + "attn.to_k.bias", p + "attn.to_v.bias"]),
                cwqkv=cat([p + "attn.add_q_proj.weight", p + "attn.add_k_proj.weight", p + "attn.add_v_proj.weight"]),
                cbqkv=cat([p + "attn.add_q_proj.bias", p + "attn.add_k_proj.bias", p + "attn.add_v_proj.bias"]),
                nq=g(p + "attn.norm_q.weight"), nk=g(p + "attn.norm_k.weight"),
                cnq=g(p + "attn.norm_added_q.weight"), cnk=g(p + "attn.norm_added_k.weight"),
                wo=g(p + "attn.to_out.0.weight"), bo=g(p + "attn.to_out.0.bias"),
                cwo=g(p + "attn.to_add_out.weight"), cbo=g(p + "attn.to_add_out.bias"),
                w1=g(p + "ff.net.0.proj.weight"), b1=g(p + "ff.net.0.proj.bias"),
                w2=g(p + "ff.net.2.weight"), b2=g(p + "ff.net.2.bias"),
                cw1=g(p + "ff_context.net.0.proj.weight"), cb1=g(p + "ff_context.net.0.proj.bias"),
                cw2=g(p + "ff_context.net.2.weight"), cb2=g(p + "ff_context.net.2.bias")))
        self.single = []
        for i in range(cfg.num_single_layers):
            p = f"single_transformer_blocks.{i}."
            # to_q | to_k | to_v | proj_mlp stacked in one weight: either two GEMMs over its row ranges (default) or ONE two-destination
            # launch ($DRAG_QKV_MLP_FUSED=1: q/k/v into the qkv buffer, the GELU'd MLP hidden into the [attn | mlp] buffer)
            self.single.append(dict(
                wqkvm=cat([p + "attn.to_q.weight", p + "attn.to_k.weight", p + "attn.to_v.weight", p + "proj_mlp.weight"]),
                bqkvm=cat([p + "attn.to_q.bias", p + "attn.to_k.bias", p + "attn.to_v.bias", p + "proj_mlp.bias"]),
                nq=g(p + "attn.norm_q.weight"), nk=g(p + "attn.norm_k.weight"),
                wo=g(p + "proj_out.weight"), bo=g(p + "proj_out.bias")))
        self._ws_key = None
        self._rope_key = None

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, B, St, Si):
        key = (B, St, Si)
        if self._ws_key == key:
            return self._ws
        cfg, dev = self.cfg, self.device
        D, S, H = cfg.dim, St + Si, cfg.num_attention_heads
        F = cfg.mlp_ratio * D
        bf = dict(dtype=torch.bfloat16, device=dev)
        s_pad = (S + 63) // 64 * 64
        ws = dict(
            x=torch.empty((B, S, D), **bf),
            nrm=torch.empty((B * S, D), **bf),
            qkv=torch.empty((B, S, 3 * D), **bf),
            vt=torch.empty((B, H, 128, s_pad), **bf),
            attn=torch.empty((B, S, D), **bf),
            cat=torch.empty((B * S, D + F), **bf),      # single: [attn | mlp]; double: MLP hidden (prefix)
            mod=torch.empty((B, self.mod_total), **bf),
            out=torch.empty((B * Si, cfg.out_channels), **bf),
        )
        self._ws_key, self._ws = key, ws
        return ws

    @staticmethod
    def _ids_key(txt_ids, img_ids):
        """identity of the position ids (order-sensitive: 4 x 6 and 6 x 4 latents have the same token count and sums)"""
        return (tuple(txt_ids.shape), tuple(img_ids.shape), hash(txt_ids.cpu().float().contiguous().numpy().tobytes()),
                hash(img_ids.cpu().float().contiguous().numpy().tobytes()))

    def _rope(self, txt_ids, img_ids):
        key = self._ids_key(txt_ids, img_ids)
        if self._rope_key != key:
            cos, sin = rope_tables(torch.cat([txt_ids.cpu().float(), img_ids.cpu().float()], dim=0), self.cfg.axes_dims_rope)
            self._rope_cs = (cos.to(self.device), sin.to(self.device))
            self._rope_key = key
        return self._rope_cs

    # ------------------------------------------------------------------ pieces
    def _set_times(self, timestep, guidance):
        """host fp32 [B] sigma / guidance scale -> persistent device buffers holding bf16(bf16(x) * 1000) as fp32
        (diffusers: ``timestep.to(hidden_states.dtype) * 1000``)."""
        B = timestep.numel()
        if getattr(self, "_t1000", None) is None or self._t1000.numel() != B:
            self._t1000 = torch.empty(B, dtype=torch.float32, device=self.device)
            self._g1000 = torch.empty(B, dtype=torch.float32, device=self.device)
        self._t1000.copy_((timestep.to(torch.bfloat16) * 1000).float())
        if guidance is not None:
            self._g1000.copy_((guidance.to(torch.bfloat16) * 1000).float())

    def _temb(self, pooled, use_guidance: bool):
        """CombinedTimestep(Guidance)TextProjEmbeddings on the times held in the device buffers (see _set_times)."""
        w = self.w
        tp = ops.timestep_embedding(self._t1000, 256)
        pre = "time_text_embed."
        h = ops.gemm(tp, w[pre + "timestep_embedder.linear_1.weight"], bias=w[pre + "timestep_embedder.linear_1.bias"], act=ops.ACT_SILU)
        emb = ops.gemm(h, w[pre + "timestep_embedder.linear_2.weight"], bias=w[pre + "timestep_embedder.linear_2.bias"])
        if use_guidance:
            gp = ops.timestep_embedding(self._g1000, 256)
            h = ops.gemm(gp, w[pre + "guidance_embedder.linear_1.weight"], bias=w[pre + "guidance_embedder.linear_1.bias"], act=ops.ACT_SILU)
            emb = ops.gemm(h, w[pre + "guidance_embedder.linear_2.weight"], bias=w[pre + "guidance_embedder.linear_2.bias"], resid=emb)
        h = ops.gemm(pooled, w[pre + "text_embedder.linear_1.weight"], bias=w[pre + "text_embedder.linear_1.bias"], act=ops.ACT_SILU)
        return ops.gemm(h, w[pre + "text_embedder.linear_2.weight"], bias=w[pre + "text_embedder.linear_2.bias"], resid=emb)

    def forward(self, hidden, enc, pooled, timestep, img_ids, txt_ids, guidance=None, taps: dict | None = None):
        """hidden bf16 [B, S_img, in]; enc bf16 [B, S_txt, J]; pooled bf16 [B, P]; timestep / guidance
        fp32 [B] host or device; returns bf16 [B, S_img, out] (a view of an internal workspace)."""
        cfg = self.cfg
        D, H = cfg.dim, cfg.num_attention_heads
        F = cfg.mlp_ratio * D
        B, Si, _ = hidden.shape
        St = enc.shape[1]
        S = St + Si
        ws = self._workspace(B, St, Si)
        x, nrm, qkv, vt, attn, catb, mod = ws["x"], ws["nrm"], ws["qkv"], ws["vt"], ws["attn"], ws["cat"], ws["mod"]
        cos, sin = self._rope(txt_ids, img_ids)
        hidden = hidden.contiguous(); enc = enc.contiguous(); pooled = pooled.contiguous()
        Mi, Mt, M = B * Si, B * St, B * S
        x_img = x.view(-1)[St * D:]          # joint buffer, image rows of batch 0 onwards
        scale = 1.0 / math.sqrt(cfg.attention_head_dim)

        # embedders write straight into the joint buffer
        ops.gemm(hidden.view(Mi, -1), self.w["x_embedder.weight"], out=x_img, bias=self.w["x_embedder.bias"], M=Mi,
                 c_rows_per_batch=Si, c_batch_stride=S * D, ldc=D)
        ops.gemm(enc.view(Mt, -1), self.w["context_embedder.weight"], out=x, bias=self.w["context_embedder.bias"], M=Mt,
                 c_rows_per_batch=St, c_batch_stride=S * D, ldc=D)
        if cfg.guidance_embeds and guidance is None:
            raise ValueError("this model has a guidance embedding: pass guidance")
        if timestep is not None:      # eager call: refresh the device-side times (a replayed graph gets them from forward_graphed)
            self._set_times(torch.as_tensor(timestep, dtype=torch.float32).cpu().reshape(-1),
                            None if guidance is None else torch.as_tensor(guidance, dtype=torch.float32).cpu().reshape(-1))
        temb = self._temb(pooled, cfg.guidance_embeds)
        st = ops.act(temb, ops.ACT_SILU)
        ops.gemm(st, self.mod_w, out=mod, bias=self.mod_b)          # every block's modulation in one GEMM
        modv = mod.view(-1)
        LM = self.mod_total
        if taps is not None:
            taps["temb"] = temb.clone()
            taps["x_embed"] = x[:, St:].clone(); taps["ctx_embed"] = x[:, :St].clone()

        nrm_img = nrm.view(-1)[: Mi * D]
        nrm_txt = nrm.view(-1)[Mi * D: (Mi + Mt) * D]
        qkv_img = qkv.view(-1)[St * 3 * D:]
        attn_img = attn.view(-1)[St * D:]
        hid = catb.view(-1)                   # MLP hidden scratch for double blocks: image rows, then text rows
        hid_txt = hid[Mi * F:]

        for i, blk in enumerate(self.double):
            mo, cmo = self.mod_off[(i, "norm1")], self.mod_off[(i, "norm1_context")]
            # chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
            ops.layernorm(x_img, nrm_img, Mi, D, scale=modv[mo + D:], shift=modv[mo:], ldx=D, rows_per_batch=Si,
                          x_batch_stride=S * D, ld_mod=LM)
            ops.layernorm(x, nrm_txt, Mt, D, scale=modv[cmo + D:], shift=modv[cmo:], ldx=D, rows_per_batch=St,
                          x_batch_stride=S * D, ld_mod=LM)
            # the image-stream and text-stream Linears of a pair share N and K: one launch when they are small alone (ops.gemm_pair)
            ops.gemm_pair(dict(a=nrm_img, w=blk["wqkv"], out=qkv_img, bias=blk["bqkv"], M=Mi, lda=D, c_rows_per_batch=Si,
                               c_batch_stride=S * 3 * D, ldc=3 * D),
                          dict(a=nrm_txt, w=blk["cwqkv"], out=qkv, bias=blk["cbqkv"], M=Mt, lda=D, c_rows_per_batch=St,
                               c_batch_stride=S * 3 * D, ldc=3 * D))
            if _SEPARATE_QPREP:     # A/B switch: the round-1 route (q prepared by the pass, plain attention)
                ops.qk_norm_rope_vt(qkv, vt, blk["cnq"], blk["cnk"], blk["nq"], blk["nk"], cos, sin, B, S, H, 3 * D, St)
                ops.attention(qkv, qkv.view(-1)[D:], vt, attn, B, S, H, 3 * D, S * 3 * D, D, S * D, scale)
            else:
                # k: RMSNorm + RoPE in place, V -> V^T; q: norm_q / norm_added_q + RoPE inside the attention kernel's Q load
                ops.k_norm_rope_vt(qkv, vt, blk["cnk"], blk["nk"], cos, sin, B, S, H, 3 * D, St)
                ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, attn, B, S, H, 3 * D, S * 3 * D, D, S * D, scale,
                                    blk["cnq"], blk["nq"], cos, sin, St)
            ops.gemm_pair(dict(a=attn_img, w=blk["wo"], out=x_img, bias=blk["bo"], M=Mi, a_rows_per_batch=Si, a_batch_stride=S * D,
                               lda=D, c_rows_per_batch=Si, c_batch_stride=S * D, ldc=D, gate=modv[mo + 2 * D:], resid=x_img, ldg=LM),
                          dict(a=attn, w=blk["cwo"], out=x, bias=blk["cbo"], M=Mt, a_rows_per_batch=St, a_batch_stride=S * D,
                               lda=D, c_rows_per_batch=St, c_batch_stride=S * D, ldc=D, gate=modv[cmo + 2 * D:], resid=x, ldg=LM))
            # MLPs (the two streams are independent: both LayerNorms, both up-projections, both down-projections)
            ops.layernorm(x_img, nrm_img, Mi, D, scale=modv[mo + 4 * D:], shift=modv[mo + 3 * D:], ldx=D,
                          rows_per_batch=Si, x_batch_stride=S * D, ld_mod=LM)
            ops.layernorm(x, nrm_txt, Mt, D, scale=modv[cmo + 4 * D:], shift=modv[cmo + 3 * D:], ldx=D,
                          rows_per_batch=St, x_batch_stride=S * D, ld_mod=LM)
            ops.gemm_pair(dict(a=nrm_img, w=blk["w1"], out=hid, bias=blk["b1"], act=ops.ACT_GELU_TANH, M=Mi, lda=D, ldc=F),
                          dict(a=nrm_txt, w=blk["cw1"], out=hid_txt, bias=blk["cb1"], act=ops.ACT_GELU_TANH, M=Mt, lda=D, ldc=F))
            ops.gemm_pair(dict(a=hid, w=blk["w2"], out=x_img, bias=blk["b2"], M=Mi, lda=F, c_rows_per_batch=Si,
                               c_batch_stride=S * D, ldc=D, gate=modv[mo + 5 * D:], resid=x_img, ldg=LM),
                          dict(a=hid_txt, w=blk["cw2"], out=x, bias=blk["cb2"], M=Mt, lda=F, c_rows_per_batch=St,
                               c_batch_stride=S * D, ldc=D, gate=modv[cmo + 5 * D:], resid=x, ldg=LM))
            if taps is not None:
                taps[f"double.{i}"] = x.clone()

        cat_mlp = catb.view(-1)[D:]
        fused_qkv_mlp = (3 * D) % 256 == 0 and (_FUSED_QKV_MLP if _FUSED_QKV_MLP is not None else
                                                ops.gemm_cost(M, 3 * D + F, D) * 10 <= (ops.gemm_cost(M, 3 * D, D) + ops.gemm_cost(M, F, D)) * 9)
        for i, blk in enumerate(self.single):
            mo = self.mod_off[("s", i)]      # shift, scale, gate
            ops.layernorm(x, nrm, M, D, scale=modv[mo + D:], shift=modv[mo:], ldx=D, ld_mod=LM, rows_per_batch=S,
                          x_batch_stride=S * D)
            if fused_qkv_mlp:
                ops.gemm(nrm, blk["wqkvm"], out=qkv, bias=blk["bqkvm"], act=ops.ACT_GELU_TANH, act_n0=3 * D, M=M, lda=D, ldc=3 * D,
                         out2=cat_mlp, ldc2=D + F, n_split=3 * D)
            else:       # (test-sized widths whose q|k|v block does not end on a tile boundary)
                ops.gemm(nrm, blk["wqkvm"][: 3 * D], out=qkv, bias=blk["bqkvm"][: 3 * D], M=M, lda=D, ldc=3 * D)
                ops.gemm(nrm, blk["wqkvm"][3 * D:], out=cat_mlp, bias=blk["bqkvm"][3 * D:], act=ops.ACT_GELU_TANH, M=M, lda=D, ldc=D + F)
            if _SEPARATE_QPREP:
                ops.qk_norm_rope_vt(qkv, vt, blk["nq"], blk["nk"], blk["nq"], blk["nk"], cos, sin, B, S, H, 3 * D, 0)
                ops.attention(qkv, qkv.view(-1)[D:], vt, catb, B, S, H, 3 * D, S * 3 * D, D + F, S * (D + F), scale)
            else:
                ops.k_norm_rope_vt(qkv, vt, blk["nk"], blk["nk"], cos, sin, B, S, H, 3 * D, 0)
                ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, catb, B, S, H, 3 * D, S * 3 * D, D + F, S * (D + F), scale,
                                    blk["nq"], blk["nq"], cos, sin, 0)
            ops.gemm(catb, blk["wo"], out=x, bias=blk["bo"], M=M, lda=D + F, c_rows_per_batch=S, c_batch_stride=S * D,
                     ldc=D, gate=modv[mo + 2 * D:], resid=x, ldg=LM)
            if taps is not None:
                taps[f"single.{i}"] = x.clone()

        mo = self.mod_off["out"]             # AdaLayerNormContinuous: scale, shift
        ops.layernorm(x_img, nrm_img, Mi, D, scale=modv[mo:], shift=modv[mo + D:], ldx=D, rows_per_batch=Si,
                      x_batch_stride=S * D, ld_mod=LM)
        ops.gemm(nrm_img, self.w["proj_out.weight"], out=ws["out"], bias=self.w["proj_out.bias"], M=Mi, lda=D,
                 ldc=cfg.out_channels)
        return ws["out"].view(B, Si, cfg.out_channels)

    # ------------------------------------------------------------------ hipGraph replay
    MAX_GRAPHS = 3          # captured forwards kept alive (each owns a workspace: ~1 GB per image at 1024^2)

    def forward_graphed(self, hidden, enc, pooled, timestep, img_ids, txt_ids, guidance=None):
        """Same result as ``forward`` (bit-identical: same kernels, same order), but the ~800 launches of one forward
        are captured ONCE into a hipGraph and replayed; only the timestep / guidance values change between replays and
        travel through device buffers updated before the launch.  The GPU was never launch-bound (host enqueue 4.8 ms
        vs 83.7 ms of GPU time at B = 1): the replay (0.2 ms) frees the host thread for image I/O.
        A captured graph bakes in the addresses of everything the forward touched, so a cache entry OWNS its workspace,
        RoPE tables and time buffers (they must not be freed or reused while the graph lives), is keyed on the input
        addresses, the shapes AND the RoPE table identity (the same token count can be a different h x w), and the
        cache is a small LRU (image sizes vary from sample to sample in stage 3)."""
        t = torch.as_tensor(timestep, dtype=torch.float32).cpu().reshape(-1)
        g = None if guidance is None else torch.as_tensor(guidance, dtype=torch.float32).cpu().reshape(-1)
        rope_key = self._ids_key(txt_ids, img_ids)
        key = (hidden.data_ptr(), enc.data_ptr(), pooled.data_ptr(), tuple(hidden.shape), tuple(enc.shape), tuple(pooled.shape),
               guidance is None, rope_key)
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.pop(key, None)
        if ent is None:
            while len(cache) >= self.MAX_GRAPHS:
                cache.pop(next(iter(cache)))                 # least recently used: drops its graph, workspace and tables
            # private buffers for this capture: nothing the eager path (or another graph) will ever reallocate
            self._ws_key = self._rope_key = None
            self._t1000 = self._g1000 = None
            self._set_times(t, g)
            rope = self._rope(txt_ids, img_ids)              # host-side table build + upload happen outside capture
            ws = self._workspace(hidden.shape[0], enc.shape[1], hidden.shape[1])
            self.forward(hidden, enc, pooled, None, img_ids, txt_ids, g)   # warm-up: allocates every temporary
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(hidden, enc, pooled, None, img_ids, txt_ids, g)
            ent = dict(graph=graph, out=out, ws=ws, rope=rope, times=(self._t1000, self._g1000))
            # hand the buffers over: the next eager forward / capture allocates its own
            self._ws_key = self._rope_key = None
            self._t1000 = self._g1000 = None
        cache[key] = ent                                     # (re)insert as most recently used
        t1000, g1000 = ent["times"]
        t1000.copy_((t.to(torch.bfloat16) * 1000).float())
        if g is not None:
            g1000.copy_((g.to(torch.bfloat16) * 1000).float())
        ent["graph"].replay()
        return ent["out"]

    __call__ = forward
