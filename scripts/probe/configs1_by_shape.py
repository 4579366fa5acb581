"""configs[1]: GEMM time by launch shape (events on the launch stream), most expensive first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops, vae as vae_mod
from domain_rag_amd.engine import FluxTxt2ImgHIP, generator_noise, pack_noise
from domain_rag_amd.flux import FluxTransformerHIP
from domain_rag_amd.flux_params import FluxConfig, init_params
dev = torch.device("cuda:0")
cfg = FluxConfig(in_channels=64, guidance_embeds=False)
tr = FluxTransformerHIP(cfg, init_params(cfg, seed=0, device=dev), dev)
vcfg = vae_mod.VaeConfig()
pipe = FluxTxt2ImgHIP(tr, vae_mod.FluxVaeHIP(vcfg, vae_mod.init_params(vcfg, seed=1, device=dev), dev))
g = torch.Generator(device=dev).manual_seed(2)
pe = torch.randn(1, 512, 4096, device=dev, generator=g).bfloat16(); pp = torch.randn(1, 768, device=dev, generator=g).bfloat16()
noise = pack_noise(generator_noise(0, 1, 512, 512, 1)[0])
run = lambda: pipe(pe, pp, height=512, width=512, guidance_scale=0.0, num_inference_steps=4, noise_tokens=noise)
pipe.use_graph = False
run(); run()
rec = ops.GemmRecorder(); ops.set_recorder(rec)
run()
ops.set_recorder(None)
tot = 0.0
for (shape, n, ms, tf) in rec.by_shape()[:16]:
    tot += ms
    print(f"{str(shape):>24} x{n:4d}  {ms:7.2f} ms  {ms / n * 1e3:7.1f} us each  {tf:6.0f} TFLOP/s")
print("GEMM launches in all:", sum(r[2] for r in rec.by_shape()), "ms;", {k: (v[0], round(v[1], 1)) for k, v in rec.by_kernel().items()})
