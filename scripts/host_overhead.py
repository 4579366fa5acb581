"""host enqueue time vs GPU time of one DiT forward, eager vs hipGraph replay"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd.flux import FluxTransformerHIP, latent_image_ids
from domain_rag_amd.flux_params import FluxConfig, init_params
dev = torch.device("cuda:0")
cfg = FluxConfig.flux_fill()
m = FluxTransformerHIP(cfg, init_params(cfg, seed=0, device=dev), dev)
for B in (1, 8):
    g = torch.Generator(device=dev).manual_seed(1)
    hidden = torch.randn(B, 4096, 384, generator=g, device=dev).bfloat16()
    enc = torch.randn(B, 1241, 4096, generator=g, device=dev).bfloat16()
    pooled = torch.randn(B, 768, generator=g, device=dev).bfloat16()
    t, gd = torch.full((B,), 0.5), torch.full((B,), 30.0)
    ii, ti = latent_image_ids(64, 64), torch.zeros(1241, 3)
    for name, fn in (("eager", m.forward), ("graph", m.forward_graphed)):
        for _ in range(2): fn(hidden, enc, pooled, t, ii, ti, gd)
        torch.cuda.synchronize()
        # one forward from an idle queue (no back-pressure from a full HIP queue in the host number)
        enq = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(hidden, enc, pooled, t, ii, ti, gd); enq.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn(hidden, enc, pooled, t, ii, ti, gd)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"B={B} {name}: host enqueue {1e3*min(enq):.1f} ms/forward (idle queue), GPU complete {1e3*(t2-t0)/3:.1f} ms/forward", flush=True)
