"""Acceptance run against the REAL reference stack — the step that pins the generation numerics (the build
container has no diffusers / clip / faiss and no checkpoints, so every generation-path oracle in ``oracle/`` is "parity unpinned").

A maintainer who has the reference's environment (``pip install -r requirements.txt``: diffusers 0.33.1, transformers, openai
clip, faiss) and its checkpoints (``./model/FLUX.1-dev``, ``FLUX.1-Fill-dev``, ``FLUX.1-Redux-dev``; ``~/.cache/clip/ViT-B-32.pt``)
runs, on an MI355X box, from the reference checkout's root:

    python /path/to/repo/scripts/accept_real_weights.py --model-root ./model --coco-dir ./retrieval/coco/train2017 \
        --target ./lamainpaint/ArTaxOr/1_shot/<sample>.jpg --out accept_report.json

It replays the reference's own call sequences with the reference's own libraries on identical inputs and seeds

    retrieval : clip.load("ViT-B/32") -> encode_image -> x / x.norm -> faiss.IndexFlatIP.add / .search(q, 100)
                                                             (retrieval/clip100_resnet_style_all_shots.py:209,171-172,425-434)
    stage 2   : pipe_prior_redux([coco, target], prompt, scales [0.8, 1.0]) -> pipe(guidance 3.5, n steps, 1024^2, generator)
                                                             (batch_generate_flux_kshot.py:459-474)
    stage 3   : pipe_prior_redux([bg]) -> pipe_fill(image, mask_image, h, w, guidance 30, n steps, generator, strength)
                                                             (outpainting_updown_sampling_redux.py:1237-1257)

then runs this repository's HIP path (``domain_rag_amd.compat`` classes — the drop-in boundary) on the same inputs, and writes
per-step latent deltas, final-pixel deltas and top-100 index equality to a JSON report with a pass/fail verdict against
BASELINE.json's bars: top-k indices bit-exact, pixels within 1e-2 (relative to full scale) of the reference on identical seeds.

Before any arithmetic it compares the KEY SETS and shapes of the real checkpoint files (safetensors headers of the transformer, VAE,
SigLIP image encoder and Redux embedder directories) with what this repository's loaders expect, and stops there if they differ.

Exit codes: 0 all requested sections pass; 1 a section fails; 2 the reference stack / checkpoints are not available here (the
report lists what is missing — that is the expected outcome inside the build container).

The reference side runs on ``--ref-device`` (cpu: the reference's CPU path, slow but device-independent; cuda: diffusers on
PyTorch-ROCm).  Nothing in this script is on the product path.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PIXEL_REL_TOL = 1e-2          # BASELINE.json north_star: "pixels within 1e-2 rel fp16 of reference"
LATENT_REL_TOL = 2e-2         # max |d| / max |ref| per step, bf16 pipelines on both sides


# --------------------------------------------------------------------------------------------------- pure comparisons
def compare_topk(D_ref, I_ref, D_hip, I_hip) -> dict:
    """index equality (the bar), plus where and how it differs when it does: a swap between two candidates whose reference
    scores tie to within the embedding error is reported separately from a genuinely different neighbour"""
    I_ref, I_hip = np.asarray(I_ref), np.asarray(I_hip)
    D_ref, D_hip = np.asarray(D_ref, np.float64), np.asarray(D_hip, np.float64)
    same = I_ref == I_hip
    out = {"queries": int(I_ref.shape[0]), "k": int(I_ref.shape[1]), "indices_equal": bool(same.all()),
           "rows_equal": int(same.all(axis=1).sum()), "max_abs_score_delta": float(np.abs(D_ref - D_hip).max())}
    if not same.all():
        q, r = np.argwhere(~same)[0]
        out["first_mismatch"] = {"query": int(q), "rank": int(r), "ref_index": int(I_ref[q, r]), "hip_index": int(I_hip[q, r]),
                                 "ref_score": float(D_ref[q, r]), "hip_score": float(D_hip[q, r])}
        out["same_sets"] = int(sum(set(a) == set(b) for a, b in zip(I_ref.tolist(), I_hip.tolist())))
    return out


def compare_latents(ref_steps, hip_steps) -> dict:
    """ref_steps / hip_steps: lists of arrays [B, n_tok, 64], one per denoise step (callback_on_step_end latents)"""
    n = min(len(ref_steps), len(hip_steps))
    rel = []
    for a, b in zip(ref_steps[:n], hip_steps[:n]):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        rel.append(float(np.abs(a - b).max() / (np.abs(a).max() + 1e-12)))
    return {"steps_compared": n, "steps_ref": len(ref_steps), "steps_hip": len(hip_steps), "rel_max_per_step": rel,
            "worst": max(rel) if rel else None, "pass": bool(rel) and len(ref_steps) == len(hip_steps) and max(rel) <= LATENT_REL_TOL}


def compare_pixels(ref_u8, hip_u8) -> dict:
    a, b = np.asarray(ref_u8, np.int32), np.asarray(hip_u8, np.int32)
    if a.shape != b.shape:
        return {"pass": False, "shape_ref": list(a.shape), "shape_hip": list(b.shape)}
    d = np.abs(a - b)
    lim = PIXEL_REL_TOL * 255.0
    return {"shape": list(a.shape), "max_abs_levels": int(d.max()), "mean_abs_levels": float(d.mean()),
            "frac_over_tol": float((d > lim).mean()), "tol_levels": lim, "psnr_db": float(10 * np.log10(255.0 ** 2 / max((d.astype(np.float64) ** 2).mean(), 1e-12))),
            "pass": bool(d.max() <= np.ceil(lim))}


# --------------------------------------------------------------------------------------------------- availability
def probe(model_root: str, sections) -> dict:
    missing = []
    need_mod = {"retrieval": ("clip", "faiss"), "stage2": ("diffusers", "transformers"), "stage3": ("diffusers", "transformers")}
    need_dir = {"stage2": ("FLUX.1-dev", "FLUX.1-Redux-dev"), "stage3": ("FLUX.1-Fill-dev", "FLUX.1-Redux-dev")}
    for sec in sections:
        for m in need_mod[sec]:
            try:
                mod = importlib.import_module(m)
                if getattr(mod, "__file__", None) is None and m in ("clip", "faiss", "diffusers"):
                    missing.append(f"module {m} is a stand-in, not the real package")
            except Exception as e:  # noqa: BLE001
                missing.append(f"module {m}: {type(e).__name__}: {e}")
        for dname in need_dir.get(sec, ()):
            if not os.path.isdir(os.path.join(model_root, dname)):
                missing.append(f"checkpoint directory {os.path.join(model_root, dname)}")
    clip_pt = os.environ.get("DRAG_CLIP_WEIGHTS") or os.path.expanduser("~/.cache/clip/ViT-B-32.pt")
    if "retrieval" in sections and not os.path.isfile(clip_pt):
        missing.append(f"CLIP checkpoint {clip_pt} (openai-CLIP's download cache, or $DRAG_CLIP_WEIGHTS)")
    return {"missing": sorted(set(missing)), "clip_checkpoint": clip_pt}


# --------------------------------------------------------------------------------------------------- checkpoint layout
def compare_key_sets(found: dict, expected: dict, ignore_prefixes=()) -> dict:
    """found / expected: tensor name -> shape.  Equality of the KEY SETS and of every shape is the first thing checked against
    real files, before any arithmetic: a renamed, missing or transposed tensor is a loading bug, not a numerics delta."""
    fk = {k for k in found if not any(k.startswith(pre) for pre in ignore_prefixes)}
    ek = set(expected)
    missing, unexpected = sorted(ek - fk), sorted(fk - ek)
    shapes = {k: {"file": list(found[k]), "expected": list(expected[k])} for k in sorted(ek & fk) if tuple(found[k]) != tuple(expected[k])}
    return {"tensors_in_file": len(fk), "tensors_expected": len(ek), "missing_from_file": missing[:20], "n_missing": len(missing),
            "unexpected_in_file": unexpected[:20], "n_unexpected": len(unexpected), "shape_mismatch": dict(list(shapes.items())[:20]),
            "n_shape_mismatch": len(shapes), "pass": not missing and not unexpected and not shapes}


def safetensors_dir_shapes(path: str) -> dict:
    """tensor name -> shape of every *.safetensors shard under ``path``, from the file headers alone (no tensor is read)"""
    from safetensors import safe_open
    out = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            with safe_open(os.path.join(path, f), framework="pt") as sf:
                for k in sf.keys():
                    out[k] = tuple(sf.get_slice(k).get_shape())
    if not out:
        raise FileNotFoundError(f"no .safetensors under {path}")
    return out


def expected_siglip_shapes(cfg) -> dict:
    """transformers SiglipVisionModel names (what FLUX.1-Redux-dev/image_encoder holds and vit.siglip_to_generic consumes)"""
    D, F, L, T = cfg.hidden, cfg.intermediate, cfg.layers, cfg.tokens
    p = "vision_model."
    s = {p + "embeddings.patch_embedding.weight": (D, 3, cfg.patch_size, cfg.patch_size), p + "embeddings.patch_embedding.bias": (D,),
         p + "embeddings.position_embedding.weight": (T, D), p + "post_layernorm.weight": (D,), p + "post_layernorm.bias": (D,)}
    for i in range(L):
        b = f"{p}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            s[b + n + ".weight"], s[b + n + ".bias"] = (D,), (D,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[b + "self_attn." + n + ".weight"], s[b + "self_attn." + n + ".bias"] = (D, D), (D,)
        s[b + "mlp.fc1.weight"], s[b + "mlp.fc1.bias"] = (F, D), (F,)
        s[b + "mlp.fc2.weight"], s[b + "mlp.fc2.bias"] = (D, F), (D,)
    return s


def section_checkpoints(args, info) -> dict:
    """key-set + shape equality between the real checkpoint files and what this repository's loaders expect
    (batch_generate_flux_kshot.py:117-153, outpainting_updown_sampling_redux.py:500-543 load the same directories)"""
    from domain_rag_amd import flux_params, redux, vae, vit
    out, ok = {}, True
    sections = info.get("sections", ())
    kinds = [k for k, sec in (("FLUX.1-dev", "stage2"), ("FLUX.1-Fill-dev", "stage3")) if sec in sections]
    for d in kinds:
        root = os.path.join(args.model_root, d)
        cfg = flux_params.FluxConfig.from_json(os.path.join(root, "transformer", "config.json"))
        out[d + "/transformer"] = compare_key_sets(safetensors_dir_shapes(os.path.join(root, "transformer")), flux_params.param_shapes(cfg))
        # the VAE files also carry the encoder's / decoder's mid-block attention under legacy names in some exports and the
        # (unused here) quant convs: everything the loader needs must be there with the right shape, extras are listed
        out[d + "/vae"] = compare_key_sets(safetensors_dir_shapes(os.path.join(root, "vae")), vae.param_shapes(vae.VaeConfig()),
                                           ignore_prefixes=("quant_conv.", "post_quant_conv."))
    if kinds:
        root = os.path.join(args.model_root, "FLUX.1-Redux-dev")
        vcfg = vit.VitConfig.siglip_so400m()
        # SiglipVisionModel's pooling head is in the file and not on the path (last_hidden_state is what Redux consumes)
        out["FLUX.1-Redux-dev/image_encoder"] = compare_key_sets(safetensors_dir_shapes(os.path.join(root, "image_encoder")),
                                                                 expected_siglip_shapes(vcfg), ignore_prefixes=("vision_model.head.",))
        out["FLUX.1-Redux-dev/image_embedder"] = compare_key_sets(safetensors_dir_shapes(os.path.join(root, "image_embedder")),
                                                                  redux.param_shapes(vcfg.hidden, 4096))
    for v in out.values():
        ok = ok and v["pass"]
    out["pass"] = ok
    return out


# --------------------------------------------------------------------------------------------------- sections
def _load_rgb(path, size=None):
    from PIL import Image
    im = Image.open(path).convert("RGB")
    return im.resize(size, Image.BICUBIC) if size else im


def section_retrieval(args, info) -> dict:
    import torch
    import clip                      # the real openai package
    import faiss
    from PIL import Image
    from domain_rag_amd import retrieval as R
    files = sorted(f for f in os.listdir(args.coco_dir) if f.lower().endswith((".jpg", ".jpeg", ".png")))[: args.n_corpus]
    if len(files) < 101:
        raise RuntimeError(f"need more than 100 corpus images under {args.coco_dir}")
    paths = [os.path.join(args.coco_dir, f) for f in files]
    dev = args.ref_device
    model, preprocess = clip.load("ViT-B/32", device=dev)
    feats = []
    with torch.no_grad():
        for p in paths:                                         # the reference's loop shape (:270-287): one image at a time
            x = preprocess(Image.open(p).convert("RGB")).unsqueeze(0).to(dev)
            f = model.encode_image(x)
            f = f / f.norm(dim=-1, keepdim=True)
            feats.append(f.float().cpu().numpy()[0])
    ref = np.stack(feats).astype(np.float32)
    q_idx = list(range(0, len(paths), max(1, len(paths) // 16)))[:16]
    index = faiss.IndexFlatIP(ref.shape[1]); index.add(ref)
    D_ref, I_ref = index.search(ref[q_idx], 100)
    # HIP path, same files through the product's embedding route
    hmodel, hpre = R.load_clip("ViT-B/32", "cuda", weights=info["clip_checkpoint"])
    hip, valid = R.compute_corpus_features(hmodel, hpre, paths, batch=256)
    assert valid == paths, "the HIP path skipped images the reference embedded"
    hidx = R.IndexFlatIP(512, "cuda"); hidx.add(hip)
    D_hip, I_hip = hidx.search(hip[q_idx], 100)
    out = compare_topk(D_ref, I_ref, D_hip, I_hip)
    out["embedding_max_abs_delta"] = float(np.abs(ref - hip).max())
    out["embedding_note"] = ("reference tower ran in " + ("fp16 (CUDA path of openai-CLIP)" if dev != "cpu" else "fp32 (CPU path)")
                             + "; the HIP tower is fp32")
    # the ranking given the REFERENCE's embeddings must be bit-exact whatever the tower precision (top-k kernel vs faiss)
    hidx2 = R.IndexFlatIP(512, "cuda"); hidx2.add(ref)
    D2, I2 = hidx2.search(ref[q_idx], 100)
    out["topk_on_reference_embeddings"] = compare_topk(D_ref, I_ref, D2, I2)
    out["pass"] = out["indices_equal"] and out["topk_on_reference_embeddings"]["indices_equal"]
    return out


def _latent_recorder(store):
    def cb(pipe, i, t, kw):
        store.append(kw["latents"].detach().float().cpu().numpy())
        return kw
    return cb


def _ref_pipes(args, kind):
    """the reference's load_model() (batch_…:117-153 / outpainting_…:500-543) with the real diffusers / transformers"""
    import torch
    from diffusers import FluxFillPipeline, FluxPipeline, FluxPriorReduxPipeline
    from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5TokenizerFast
    dt = torch.bfloat16
    flux = os.path.join(args.model_root, "FLUX.1-Fill-dev" if kind == "fill" else "FLUX.1-dev")
    te = CLIPTextModel.from_pretrained(flux, subfolder="text_encoder", torch_dtype=dt)
    te2 = T5EncoderModel.from_pretrained(flux, subfolder="text_encoder_2", torch_dtype=dt)
    tok, tok2 = CLIPTokenizer.from_pretrained(flux, subfolder="tokenizer"), T5TokenizerFast.from_pretrained(flux, subfolder="tokenizer_2")
    prior = FluxPriorReduxPipeline.from_pretrained(os.path.join(args.model_root, "FLUX.1-Redux-dev"), text_encoder=te, text_encoder_2=te2,
                                                   tokenizer=tok, tokenizer_2=tok2, torch_dtype=dt).to(args.ref_device)
    if kind == "fill":
        pipe = FluxFillPipeline.from_pretrained(flux, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None,
                                                torch_dtype=dt).to(args.ref_device)
    else:
        pipe = FluxPipeline.from_pretrained(flux, text_encoder=None, text_encoder_2=None, torch_dtype=dt).to(args.ref_device)
    return prior, pipe, (te, te2, tok, tok2)


def _hip_pipes(args, kind, encoders):
    from domain_rag_amd import compat
    flux = os.path.join(args.model_root, "FLUX.1-Fill-dev" if kind == "fill" else "FLUX.1-dev")
    te, te2, tok, tok2 = encoders
    prior = compat.FluxPriorReduxPipeline.from_pretrained(os.path.join(args.model_root, "FLUX.1-Redux-dev"), text_encoder=te, text_encoder_2=te2,
                                                          tokenizer=tok, tokenizer_2=tok2).to("cuda")
    cls = compat.FluxFillPipeline if kind == "fill" else compat.FluxPipeline
    return prior, cls.from_pretrained(flux).to("cuda")


def _prior_delta(ref_out, hip_out) -> dict:
    r = {}
    for k in ("prompt_embeds", "pooled_prompt_embeds"):
        a = ref_out[k].detach().float().cpu().numpy().astype(np.float64)
        b = hip_out[k].detach().float().cpu().numpy().astype(np.float64)
        r[k] = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-12))
    return r


def section_stage2(args, info) -> dict:
    import torch
    coco = _load_rgb(os.path.join(args.coco_dir, sorted(os.listdir(args.coco_dir))[0]))
    target = _load_rgb(args.target)
    call = dict(prompt="", prompt_2="", prompt_embeds_scale=[0.8, 1.0], pooled_prompt_embeds_scale=[1.0, 1.0])
    ref_prior, ref_pipe, enc = _ref_pipes(args, "dev")
    ref_lat: list = []
    with torch.no_grad():
        ro = ref_prior(image=[coco, target], **call)
        ref_img = ref_pipe(guidance_scale=3.5, num_inference_steps=args.steps, height=1024, width=1024,
                           generator=torch.Generator("cpu").manual_seed(args.seed), callback_on_step_end=_latent_recorder(ref_lat),
                           **ro).images[0]
    ref_keep = {k: ro[k].detach().cpu() for k in ("prompt_embeds", "pooled_prompt_embeds")}
    del ref_prior, ref_pipe
    hip_prior, hip_pipe = _hip_pipes(args, "dev", enc)
    hip_lat: list = []
    ho = hip_prior(image=[coco, target], **call)
    hip_img = hip_pipe(guidance_scale=3.5, num_inference_steps=args.steps, height=1024, width=1024,
                       generator=torch.Generator("cpu").manual_seed(args.seed), callback_on_step_end=_latent_recorder(hip_lat), **ho).images[0]
    out = {"prior_rel_delta": _prior_delta(ref_keep, ho), "latents": compare_latents(ref_lat, hip_lat),
           "pixels": compare_pixels(np.asarray(ref_img), np.asarray(hip_img))}
    out["pass"] = out["latents"]["pass"] and out["pixels"]["pass"]
    return out


def section_stage3(args, info) -> dict:
    import torch
    from PIL import Image
    from domain_rag_amd import hostlogic as H
    image = _load_rgb(args.target, (1024, 1024))
    bg = _load_rgb(os.path.join(args.coco_dir, sorted(os.listdir(args.coco_dir))[1]))
    mask = H.generate_outpaint_mask((1024, 1024), [[362, 362, 300, 300]]) if hasattr(H, "generate_outpaint_mask") else None
    if mask is None or not isinstance(mask, Image.Image):
        arr = np.full((1024, 1024), 255, np.uint8); arr[362:663, 362:663] = 0
        mask = Image.fromarray(arr, "L")
    n, strength = args.steps_fill, args.strength
    ref_prior, ref_pipe, enc = _ref_pipes(args, "fill")
    ref_lat: list = []
    with torch.no_grad():
        ro = ref_prior(image=[bg], prompt="", prompt_2="", prompt_embeds_scale=[1.0], pooled_prompt_embeds_scale=[1.0])
        ref_img = ref_pipe(image=image, mask_image=mask, height=1024, width=1024, guidance_scale=30, num_inference_steps=n,
                           prompt_embeds=ro.prompt_embeds, pooled_prompt_embeds=ro.pooled_prompt_embeds,
                           generator=torch.Generator("cpu").manual_seed(args.seed), strength=strength,
                           callback_on_step_end=_latent_recorder(ref_lat)).images[0]
    ref_keep = {k: ro[k].detach().cpu() for k in ("prompt_embeds", "pooled_prompt_embeds")}
    del ref_prior, ref_pipe
    hip_prior, hip_pipe = _hip_pipes(args, "fill", enc)
    hip_lat: list = []
    ho = hip_prior(image=[bg], prompt="", prompt_2="", prompt_embeds_scale=[1.0], pooled_prompt_embeds_scale=[1.0])
    hip_img = hip_pipe(image=image, mask_image=mask, height=1024, width=1024, guidance_scale=30, num_inference_steps=n,
                       prompt_embeds=ho.prompt_embeds, pooled_prompt_embeds=ho.pooled_prompt_embeds,
                       generator=torch.Generator("cpu").manual_seed(args.seed), strength=strength,
                       callback_on_step_end=_latent_recorder(hip_lat)).images[0]
    out = {"prior_rel_delta": _prior_delta(ref_keep, ho), "latents": compare_latents(ref_lat, hip_lat),
           "pixels": compare_pixels(np.asarray(ref_img), np.asarray(hip_img)), "strength": strength, "num_inference_steps": n}
    out["pass"] = out["latents"]["pass"] and out["pixels"]["pass"]
    return out


SECTIONS = {"retrieval": section_retrieval, "stage2": section_stage2, "stage3": section_stage3}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model-root", default="./model")
    ap.add_argument("--coco-dir", default="./retrieval/coco/train2017")
    ap.add_argument("--target", default=None, help="a LaMa-inpainted k-shot image (stage 2 / 3 input)")
    ap.add_argument("--n-corpus", type=int, default=1000)
    ap.add_argument("--sections", default="retrieval,stage2,stage3")
    ap.add_argument("--ref-device", default="cpu")
    ap.add_argument("--steps", type=int, default=8, help="stage-2 denoise steps (the reference uses 50; the comparison is per step)")
    ap.add_argument("--steps-fill", type=int, default=10)
    ap.add_argument("--strength", type=float, default=0.6)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="accept_report.json")
    args = ap.parse_args(argv)
    sections = [s for s in args.sections.split(",") if s]
    bad = [s for s in sections if s not in SECTIONS]
    if bad:
        ap.error(f"unknown section(s) {bad}; choose from {sorted(SECTIONS)}")
    info = probe(args.model_root, sections)
    report = {"sections_requested": sections, "tolerances": {"pixels_rel_full_scale": PIXEL_REL_TOL, "latents_rel_max": LATENT_REL_TOL,
                                                             "topk": "indices bit-exact"}, "missing": info["missing"]}
    if any(s in sections for s in ("stage2", "stage3")) and not args.target:
        report["missing"].append("--target <inpainted k-shot image>")
    if report["missing"]:
        report["verdict"] = "not-run: the reference stack is not available here"
        with open(args.out, "w") as f:
            json.dump(report, f, indent=2)
        print(json.dumps(report, indent=2))
        return 2
    ok = True
    # checkpoint layout first: no arithmetic is compared over weights that did not load the way the reference loads them
    info["sections"] = sections
    try:
        report["checkpoints"] = section_checkpoints(args, info)
    except Exception as e:  # noqa: BLE001
        report["checkpoints"] = {"pass": False, "error": f"{type(e).__name__}: {e}"}
    if not report["checkpoints"]["pass"]:
        report["verdict"] = "fail: checkpoint key sets / shapes differ from what the loaders expect (arithmetic sections not run)"
        with open(args.out, "w") as f:
            json.dump(report, f, indent=2)
        print(json.dumps(report["checkpoints"], indent=2))
        return 1
    for s in sections:
        try:
            report[s] = SECTIONS[s](args, info)
        except Exception as e:  # noqa: BLE001
            report[s] = {"pass": False, "error": f"{type(e).__name__}: {e}"}
        ok = ok and bool(report[s].get("pass"))
    report["verdict"] = "pass" if ok else "fail"
    with open(args.out, "w") as f:
        json.dump(report, f, indent=2)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("pass", "error", "indices_equal")})
                      for k, v in report.items()}, indent=2))
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
