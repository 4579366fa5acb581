"""BASELINE configs[1] (Flux-schnell shape, 512x512, 4 steps, batch 1) alone, for rocprofv3 --kernel-trace (and per GEMM shape
with event timing): where the latency case spends its time.  REPS runs after 2 warm-ups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import vae as vae_mod, ops
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
run(); run(); torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "5"))
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"configs[1]: {dt*1e3:.1f} ms per image = {1/dt:.2f} img/s", flush=True)
if os.environ.get("BY_SHAPE"):
    # one more image with every GEMM launch bracketed by events (eager launches: the graph replay path takes no events)
    rec = ops.GemmRecorder()
    ops.set_recorder(rec)
    pipe.use_graph = False
    try:
        run()
    finally:
        ops.set_recorder(None)
    rows = rec.by_shape()
    tot = sum(r[2] for r in rows)
    print(f"{'M':>8} {'N':>8} {'K':>6} {'launches':>8} {'total_ms':>10} {'pct':>6} {'TF/s':>8}")
    for (M, N, K), n, ms, tf in rows[:24]:
        print(f"{M:8d} {N:8d} {K:6d} {n:8d} {ms:10.2f} {100*ms/tot:6.2f} {tf:8.1f}")
    fl, ms, n = rec.totals()
    print(f"GEMM total {n} launches {ms:.1f} ms {fl/ms/1e9:.1f} TF/s")
