"""Level-2 boundary: the reference scripts' own call sequences against the stand-in modules
`compat.install()` registers, checked bit-for-bit against the direct HIP host classes (which the other GPU tests
check against the oracle)."""
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def installed(monkeypatch):
    from domain_rag_amd import compat
    monkeypatch.setenv("DRAG_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("DRAG_TINY", "1")
    saved = {k: sys.modules.get(k) for k in ("clip", "faiss", "diffusers")}
    compat.install()
    yield compat
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _pil(seed, w, h, mode="RGB"):
    from PIL import Image
    rng = np.random.default_rng(seed)
    shape = (h, w, 3) if mode == "RGB" else (h, w)
    return Image.fromarray(rng.integers(0, 256, shape, dtype=np.uint8), mode=mode)


def test_reference_call_sites_fill(gpu, installed):
    """outpainting_updown_sampling_redux.py:525-541 (load) and :1237-1257 (per background)"""
    from diffusers import FluxFillPipeline, FluxPriorReduxPipeline
    from domain_rag_amd.engine import Engine, generator_noise, pack_noise
    pipe_prior_redux = FluxPriorReduxPipeline.from_pretrained("./model/FLUX.1-Redux-dev", text_encoder=None, text_encoder_2=None,
                                                              tokenizer=None, tokenizer_2=None, torch_dtype=torch.bfloat16).to("cuda")
    pipe_fill = FluxFillPipeline.from_pretrained("./model/FLUX.1-Fill-dev", text_encoder=None, text_encoder_2=None,
                                                 tokenizer=None, tokenizer_2=None, torch_dtype=torch.bfloat16).to("cuda")
    bg, image = _pil(0, 80, 72), _pil(1, 96, 64)
    mask = _pil(2, 96, 64, "L").point(lambda v: 255 if v > 100 else 0)
    prior = pipe_prior_redux([bg], prompt="", prompt_2="", prompt_embeds_scale=[1.2], pooled_prompt_embeds_scale=[1.0])
    assert set(dict(**prior)) == {"prompt_embeds", "pooled_prompt_embeds"}
    result = pipe_fill(image=image, mask_image=mask, height=64, width=96, guidance_scale=30.0, num_inference_steps=4,
                       prompt_embeds=prior.prompt_embeds, pooled_prompt_embeds=prior.pooled_prompt_embeds,
                       generator=torch.Generator("cpu").manual_seed(1234), strength=0.8).images[0]
    assert result.size == (96, 64)
    # the same through the host classes directly
    eng = Engine("fill", synthetic=True, tiny=True)
    pe, pp = eng.prior_embeds([bg], "", [1.2], [1.0])
    assert torch.equal(pe, prior["prompt_embeds"]) and torch.equal(pp, prior["pooled_prompt_embeds"])
    en, noise, mn = generator_noise(1234, 1, 64, 96, 3)
    out = eng.pipe(torch.from_numpy(np.asarray(image).copy())[None].to(gpu), torch.from_numpy(np.asarray(mask).copy())[None].to(gpu),
                   pe, pp, guidance_scale=30.0, num_inference_steps=4, strength=0.8, enc_noise=en.to(gpu),
                   masked_enc_noise=mn.to(gpu), noise_tokens=pack_noise(noise).to(gpu))
    assert np.array_equal(np.asarray(result), out[0].cpu().numpy())
    with pytest.raises(ValueError):
        pipe_fill(image=image, mask_image=mask, height=64, width=96, num_inference_steps=4, strength=0.1, **prior)


def test_reference_call_sites_txt2img(gpu, installed):
    """batch_generate_flux_kshot.py:139-151 (load) and :459-474 (generate_image)"""
    from diffusers import FluxPipeline, FluxPriorReduxPipeline
    from domain_rag_amd.engine import Engine, generator_noise, pack_noise
    pipe_prior_redux = FluxPriorReduxPipeline.from_pretrained("./model/FLUX.1-Redux-dev", torch_dtype=torch.bfloat16).to("cuda")
    pipe = FluxPipeline.from_pretrained("./model/FLUX.1-dev", torch_dtype=torch.bfloat16).to("cuda")
    coco, target = _pil(3, 70, 90), _pil(4, 64, 64)
    prior = pipe_prior_redux([coco, target], prompt=["", ""], prompt_2=["", ""], prompt_embeds_scale=[0.8, 1.0],
                             pooled_prompt_embeds_scale=[1.0, 1.0])
    images = pipe(guidance_scale=2.5, num_inference_steps=3, height=64, width=64,
                  generator=torch.Generator("cpu").manual_seed(0), **prior).images
    assert len(images) == 1 and images[0].size == (64, 64)
    eng = Engine("dev", synthetic=True, tiny=True)
    pe, pp = eng.prior_embeds([coco, target], "", [0.8, 1.0], [1.0, 1.0])
    out = eng.pipe(pe, pp, height=64, width=64, guidance_scale=2.5, num_inference_steps=3,
                   noise_tokens=pack_noise(generator_noise(0, 1, 64, 64, 1)[0]))
    assert np.array_equal(np.asarray(images[0]), out[0].cpu().numpy())


def test_reference_call_sites_retrieval(gpu, installed):
    """retrieval/clip100_resnet_style_all_shots.py:209 (clip.load), :171 (encode_image), :425-434 (IndexFlatIP)"""
    import clip
    import faiss
    model, preprocess = clip.load("ViT-B/32", device="cuda")
    x = torch.stack([preprocess(_pil(10 + i, 300, 200)) for i in range(3)])
    with torch.no_grad():
        f = model.encode_image(x.to("cuda"))
        f = f / f.norm(dim=-1, keepdim=True)
    f = f.cpu().numpy()
    assert f.shape == (3, 512) and f.dtype == np.float32
    rng = np.random.default_rng(0)
    corpus = rng.standard_normal((500, 512)).astype(np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    corpus[123] = f[1]
    index = faiss.IndexFlatIP(corpus.shape[1])
    index.add(corpus)
    D, I = index.search(f[1:2], min(100, len(corpus)))
    assert D.shape == (1, 100) and I.dtype == np.int64 and I[0, 0] == 123 and abs(D[0, 0] - 1.0) < 1e-5
    assert np.all(np.diff(D[0]) <= 0)
    from oracle import retrieval as oret
    Do, Io = oret.cosine_topk(corpus, f[1:2], 100)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
