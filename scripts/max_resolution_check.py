"""the stage-3 resolution cap (2800 px) end to end on the HIP path: full-size VAE + SigLIP/Redux, a shallow DiT of the real
width (2 double + 2 single blocks), B=1, 2 steps — checks 32-bit offset limits, LDS-DMA descriptor spans and memory"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd.fill_pipeline import SyntheticFillJob
from domain_rag_amd.flux_params import FluxConfig
dev = torch.device("cuda:0")
W, H = 2800, 2096
class Job(SyntheticFillJob):
    pass
cfg = FluxConfig(in_channels=384, num_layers=2, num_single_layers=2)
# SyntheticFillJob is square; build the rectangular inputs by hand around its models
job = SyntheticFillJob(batch=1, res=1024, denoise_steps=2, device=dev, seed=0, cfg=cfg)
g = torch.Generator(device=dev).manual_seed(1)
image = torch.randint(0, 256, (1, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
mask = torch.full((1, H, W), 255, device=dev, dtype=torch.uint8); mask[:, 500:900, 700:1500] = 0
pe, pp = job.prior(job.bg, job.t5, job.pooled, [1.0], [1.0], group=1)
en = torch.randn((1, 16, H // 8, W // 8), generator=g, device=dev).bfloat16()
mn = torch.randn((1, 16, H // 8, W // 8), generator=g, device=dev).bfloat16()
nt = torch.randn((1, (H // 16) * (W // 16), 64), generator=g, device=dev).bfloat16()
torch.cuda.synchronize(); t0 = time.perf_counter()
out = job.fill(image, mask, pe, pp, guidance_scale=30.0, num_inference_steps=2, strength=1.0, enc_noise=en, masked_enc_noise=mn, noise_tokens=nt)
torch.cuda.synchronize()
print(f"{W}x{H}: out {tuple(out.shape)} {out.dtype}, mean {out.float().mean().item():.1f}, {time.perf_counter()-t0:.1f} s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
assert out.shape == (1, H, W, 3) and out.dtype == torch.uint8
print("MAX-RES OK")
