"""Prints the HIP-vs-oracle error levels and oracle timings at full size (sets the tolerances of tests/test_gpu_fullsize2.py).
Test infrastructure (it runs the oracle), hence under tests/: python tests/tools/explore_fullsize_errors.py on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from domain_rag_amd import vae, vit
from oracle import vae as ov, vit as ovit
torch.set_num_threads(os.cpu_count() or 1)
gpu = torch.device("cuda:0")
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-12)).item()
cfg = vae.VaeConfig(); p = vae.init_params(cfg, seed=3); p32 = {k: v.float() for k, v in p.items()}
m = vae.FluxVaeHIP(cfg, p, gpu)
for h in (16, 32, 64):
    tok = (torch.randn((1, h * h, 64), generator=torch.Generator().manual_seed(5))).bfloat16()
    img_u8, rows = m.decode_tokens(tok.to(gpu), 1, h, h, return_rows=True)
    H = img_u8.shape[1]
    got = (rows.view(1, H, H, -1)[..., :3].float().cpu() / 2 + 0.5).clamp(0, 1).permute(0, 3, 1, 2)
    with torch.no_grad():
        t0 = time.time(); _, r32 = ov.decode_tokens_to_u8(p32, tok.float(), h, h); t32 = time.time() - t0
        tb = eb = None
        if h <= 32:
            t0 = time.time(); _, rb = ov.decode_tokens_to_u8(p, tok, h, h); tb = time.time() - t0; eb = rel(rb, r32)
    d = (img_u8.cpu()[0].int() - (r32[0].permute(1, 2, 0) * 255).round().int()).abs()
    print(f"vae decode h={h}: hip-vs-fp32 {rel(got, r32):.3e}  bf16oracle-vs-fp32 {eb}  pix max {d.max().item()} frac>1 {(d>1).float().mean().item():.2e}  oracle fp32 {t32:.1f}s bf16 {tb}", flush=True)
c = vit.VitConfig.siglip_so400m(); g = vit.init_generic_params(c, 11)
img = (torch.rand(2, 384, 384, 3, generator=torch.Generator().manual_seed(2)) * 255).to(torch.uint8)
px = ovit.normalize_u8(img, c.mean, c.std)
t0 = time.time(); r32 = ovit.siglip_last_hidden_state(g, 384, 14, 1152, 16, 27, 4304, px, torch.float32); t32 = time.time() - t0
t0 = time.time(); rb = ovit.siglip_last_hidden_state(g, 384, 14, 1152, 16, 27, 4304, px, torch.bfloat16); tb = time.time() - t0
out = vit.VitHIP(c, g, gpu)(img.to(gpu))
print(f"siglip full: hip-vs-fp32 {rel(out, r32):.3e} bf16oracle-vs-fp32 {rel(rb, r32):.3e} mean-rel {((out.float().cpu()-r32).abs().mean()/r32.abs().mean()).item():.3e} oracle fp32 {t32:.1f}s bf16 {tb:.1f}s", flush=True)
