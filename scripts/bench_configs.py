"""BASELINE configs[1] (Flux-schnell shape, 512x512, 4 steps, batch 1) and the retrieval half of configs[0] on one GPU:
latency-style measurements that are not bench.py lines (DESIGN.md quotes them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import vae as vae_mod, retrieval as R, ops
from domain_rag_amd.engine import FluxTxt2ImgHIP, generator_noise, pack_noise
from domain_rag_amd.flux import FluxTransformerHIP
from domain_rag_amd.flux_params import FluxConfig, init_params
dev = torch.device("cuda:0")

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# ---- configs[1]: schnell shape (no guidance embedding), 512x512, 4 steps, B=1, text = 512 T5 tokens
cfg = FluxConfig(in_channels=64, guidance_embeds=False)
tr = FluxTransformerHIP(cfg, init_params(cfg, seed=0, device=dev), dev)
vcfg = vae_mod.VaeConfig()
pipe = FluxTxt2ImgHIP(tr, vae_mod.FluxVaeHIP(vcfg, vae_mod.init_params(vcfg, seed=1, device=dev), dev))
g = torch.Generator(device=dev).manual_seed(2)
pe = torch.randn(1, 512, 4096, device=dev, generator=g).bfloat16(); pp = torch.randn(1, 768, device=dev, generator=g).bfloat16()
noise = pack_noise(generator_noise(0, 1, 512, 512, 1)[0])
for B in (1, 8):
    pe_b, pp_b, nz = pe.expand(B, -1, -1).contiguous(), pp.expand(B, -1).contiguous(), noise.expand(B, -1, -1).contiguous()
    dt = timed(lambda: pipe(pe_b, pp_b, height=512, width=512, guidance_scale=0.0, num_inference_steps=4, noise_tokens=nz))
    S = 512 + 1024
    fl = B * (4 * (57 * (2 * 12 * 3072 * 3072 * S + 2 * 2 * S * S * 3072)) + 2.6e12)
    print(f"configs[1] schnell-shape 512x512, 4 steps, B={B}: {dt*1e3:.1f} ms per batch = {B/dt:.2f} img/s ({fl/dt/1e12:.0f} TFLOP/s)", flush=True)
del tr, pipe
torch.cuda.empty_cache()
# ---- configs[0] on the GPU: CLIP ViT-B/32 over 1000 640x480 images (decoded bytes resident) + top-100 for 16 queries
model, _ = R.load_clip("ViT-B/32", device=dev)
from domain_rag_amd import resample
raw = torch.randint(0, 256, (1000, 480, 640, 3), dtype=torch.uint8, device=dev, generator=g)
def embed_all():
    pre = resample.clip_preprocess_u8(raw)                    # same-size batch: one resize launch pair
    return R.embed_images(model, pre, 256)
dt = timed(embed_all)
feats = embed_all()
print(f"configs[0] on GPU: resize+crop+embed 1000 images {dt*1e3:.1f} ms = {1000/dt:.0f} img/s", flush=True)
q = feats[:16].contiguous()
dt = timed(lambda: ops.cosine_topk(feats, q, 100), 20)
print(f"configs[0] on GPU: top-100 of 16 queries over 1000 x 512: {dt*1e6:.0f} us", flush=True)
