"""PIL-exact resize kernels: throughput on the retrieval shapes (decoded bytes resident in HBM)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import resample
dev = torch.device("cuda:0")
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
g = torch.Generator(device=dev).manual_seed(0)
raw = torch.randint(0, 256, (1000, 480, 640, 3), dtype=torch.uint8, device=dev, generator=g)
dt = timed(lambda: resample.clip_preprocess_u8(raw))
byts = raw.numel() + 1000 * 224 * 224 * 3
print(f"CLIP resize(224, bicubic)+crop, 1000 x 640x480 in one batch: {dt*1e3:.2f} ms = {1000/dt:.0f} img/s, {byts/dt/1e9:.0f} GB/s of algorithmic bytes")
one = raw[0].contiguous()
dt1 = timed(lambda: resample.clip_preprocess_u8(one), 200)
print(f"same, one image per call: {dt1*1e6:.0f} us per image (2 launches)")
big = torch.randint(0, 256, (8, 1024, 1024, 3), dtype=torch.uint8, device=dev, generator=g)
dt = timed(lambda: resample.siglip_resize_u8(big))
print(f"SigLIP resize 384x384 bicubic, 8 x 1024x1024: {dt*1e3:.2f} ms = {8/dt:.0f} img/s, {(big.numel() + 8*384*384*3)/dt/1e9:.0f} GB/s")
