import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import vit, retrieval as R
dev = torch.device("cuda:0")
def bench(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for prec in ("fp32", "bf16"):
  model, _ = R.load_clip("ViT-B/32", dev, precision=prec)
  for B in (64, 256, 1024):
    img = torch.randint(0, 256, (B, 224, 224, 3), device=dev, dtype=torch.uint8)
    ms = bench(lambda: model.embed_normalized(img))
    print(f"CLIP ViT-B/32 {prec} embed B={B}: {ms:.2f} ms  {B/ms*1e3:.0f} img/s  {8.82e9*B/ms/1e9:.0f} TF/s (118287 images -> {118287/(B/ms*1e3):.1f} s)", flush=True)
cfg = vit.VitConfig.siglip_so400m()
sg = vit.VitHIP(cfg, vit.init_generic_params(cfg, 0, device=dev), dev)
for B in (8, 32):
    img = torch.randint(0, 256, (B, 384, 384, 3), device=dev, dtype=torch.uint8)
    ms = bench(lambda: sg(img))
    print(f"SigLIP-so400m B={B}: {ms:.2f} ms  {B/ms*1e3:.0f} img/s  {0.67e12*B/ms/1e9:.0f} TF/s", flush=True)
st = R.StemStyle(device=dev)
x = torch.rand(101, 3, 256, 256, device=dev)
ms = bench(lambda: st(x))
print(f"ResNet-stem style, 101 images (one query's re-rank set): {ms:.2f} ms", flush=True)
