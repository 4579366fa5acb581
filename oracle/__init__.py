"""oracle/ — CPU restatements of the reference's algorithm for the retrieve-then-generate hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg
may import anything from here, and only as the checker — never as the thing measured or shipped.
The product path (domain-rag_amd/) never imports this package and has no CPU fallback.

PARITY UNPINNED for the numeric core: the reference (LiYu0524/Domain-RAG) ships no tests or golden vectors, and the
wheels that hold its arithmetic (diffusers 0.33.1, transformers 4.46.3, openai/CLIP@dcba3cb,
faiss 1.10.0, torchvision 0.22.0 — requirements.txt:4,7,8,9,60,62) are neither vendored under
/root/reference nor installable offline; simple-lama-inpainting (un-pinned) and its big-lama.pt weights likewise.  Those
restatements (flux.py, vae.py, redux.py, fill.py, stem.py, topk.c, lama.py) are anchored on the reference's call sites
(cited per function).
PINNED parts: resize.py is checked bit-for-bit against PIL itself (tests/test_oracle_resize.py); vit.py drives the
`transformers` CLIP / SigLIP vision modules that ARE importable here (version 5.x, not the pinned 4.46.3) with shared
weights; stem.py's conv/bn/relu/pool equals transformers' ResNetEmbeddings (the same ResNet-50 stem) bit for bit; the host logic is checked against goldens captured from the imported reference scripts
(tests/golden/make_host_goldens.py, make_stage2_goldens.py, make_stage1_goldens.py, make_lama_goldens.py).  DESIGN.md lists the status row by row.

UPSTREAM TWINS (round 6; tests/test_oracle_upstream_twins.py — CPU, shared random weights, upstream module vs this package).
diffusers cannot be imported here, but `transformers` ships code of the same lineage with the same arithmetic:

  oracle lines                                   twin (transformers 5.x, importable here)                       bar
  vae.py  resnet()                    :27-32     janus JanusVQVAEResnetBlock (GroupNorm-32 eps 1e-6, swish,     2e-5 rel (fp32; F.silu vs x*sigmoid(x))
                                                 3x3 convs, 1x1 shortcut)
  vae.py  mid_attention()             :35-44     janus JanusVQVAEAttnBlock (single head, scale C^-0.5,          2e-5 rel (SDPA vs bmm + softmax)
                                                 residual; 1x1 convs = the Linear weights)
  vae.py  decode() up-sample          :57-58     janus JanusVQVAEConvUpsample (nearest 2x, conv pad 1)          bit-exact
  vae.py  encode_moments() down-sample:70-71     janus JanusVQVAEConvDownsample (pad (0,1,0,1), stride 2)       bit-exact
  vae.py  decode() whole stack        :47-60     janus JanusVQVAEDecoder, also at the Flux VAE's own plan       5e-5 rel
                                                 (1, 2, 4, 4) x 2 resnets (+1 in the decoder); the twin's
                                                 extra level attention silenced by a zero proj_out
  vae.py  encode_moments() whole stack:63-76     janus JanusVQVAEEncoder (double_latent: mean | logvar)         5e-5 rel
  flux.py rms_norm()                             t5 T5LayerNorm, fp32 and bf16                                  bit-exact
  flux.py apply_rope()                           gptj apply_rotary_pos_emb + rotate_every_two                   bit-exact (fp32; bf16 via the float copy)
  flux.py rope_tables() (per axis)               gptj create_sinusoidal_positions                               2e-5 abs (twin is float32, oracle float64)
  flux.py adaln_zero()  (6-chunk order)          qwen2_5_omni Qwen2_5_OmniAdaLayerNormZero, fp32 and bf16       bit-exact
  flux.py adaln_continuous() (scale, shift)      qwen2_5_omni Qwen2_5_OmniAdaLayerNormZero_Final                bit-exact
  flux.py gated_mlp_residual()                   qwen2_5_omni DiTDecoderLayer around a stand-in attention       bit-exact
  vae.py  pack_latents / unpack_latents / pack_mask  torch's own pixel_unshuffle(2) / pixel_shuffle(2) / pixel_unshuffle(8)   bit-exact
  vit.py, stem.py                                CLIP / SigLIP vision towers, ResNetEmbeddings (rounds 1-3)     see those tests

STILL REVIEW-ONLY (no importable twin; restated from diffusers 0.33.1's published code, anchored on the reference's call sites):
the joint `[txt, img]` concatenation order and the q/k RMSNorm -> RoPE -> SDPA composition inside the double / single blocks; the
single block's 3-chunk (shift, scale, gate) order and its cat([attn, mlp]) -> proj_out; `timestep_proj` (cos first, 256-d,
downscale shift 0) and the bf16-rounded x1000 timestep / guidance; `flow_sigmas` (dynamic shift, float32 table) and `euler_step`;
Fill's concatenation order [masked-image latents | mask] (fill.py; the packing index maps themselves are pinned on torch's pixel_unshuffle); the Redux concatenation (redux.py); DiagonalGaussian sampling and
the (z - shift) * scaling constants; lama.py's FFC network; topk.c's tie order (faiss defines none).
"""
