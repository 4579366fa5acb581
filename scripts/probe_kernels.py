"""GPU probe: prints device, real error magnitudes vs fp64 references and first kernel timings."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops, _lib

print("device:", torch.cuda.get_device_name(0), "| lib:", _lib.LIB_PATH, "| version", _lib.load().drag_version())
dev = torch.device("cuda:0")
with open("/proc/self/maps") as f:
    print("loaded .so:", sorted({l.split()[-1] for l in f if "domainrag" in l}))

def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

# negative control: a wrong reference must NOT match
g = torch.Generator().manual_seed(0)
a = torch.randn(512, 1024, generator=g).bfloat16(); w = (torch.randn(768, 1024, generator=g) * 0.05).bfloat16()
out = ops.gemm(a.to(dev), w.to(dev)).cpu().double()
ref = a.double() @ w.double().T
print("gemm rel err vs fp64:", ((out - ref).abs().max() / ref.abs().max()).item(),
      "| vs WRONG ref (transposed operands swapped):", ((out - (a.double().flip(0) @ w.double().T)).abs().max() / ref.abs().max()).item())

for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (32768, 3072, 3072), (42696, 9216, 3072), (42696, 12288, 3072), (42696, 3072, 15360), (32768, 3072, 12288)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm(A, W, out=C))
    print(f"gemm {M}x{N}x{K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF/s")
    del A, W, C

for (B, S, H) in [(1, 5337, 24), (8, 5337, 24)]:
    D = H * 128
    qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    wq = torch.ones(128, device=dev).bfloat16()
    cos = torch.ones(S, 64, device=dev); sin = torch.zeros(S, 64, device=dev)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    ms_p = bench(lambda: ops.qk_norm_rope_vt(qkv, vt, wq, wq, wq, wq, cos, sin, B, S, H, 3 * D, 1241))
    ms_a = bench(lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128)))
    fl = 4.0 * S * S * 128 * H * B
    print(f"attn B={B} S={S} H={H}: prep {ms_p:.3f} ms, attention {ms_a:.3f} ms  {fl/ms_a/1e9:.1f} TF/s")
    del qkv, vt, o

x = torch.randn(42696, 3072, device=dev).bfloat16(); y = torch.empty_like(x)
mod = torch.randn(8, 6 * 3072, device=dev).bfloat16()
ms = bench(lambda: ops.layernorm(x, y, 42696, 3072, scale=mod.view(-1)[3072:], shift=mod.view(-1), ldx=3072, rows_per_batch=5337, x_batch_stride=5337 * 3072, ld_mod=6 * 3072))
print(f"layernorm_modulate 42696x3072: {ms:.3f} ms  {2*x.numel()*2/ms/1e6:.0f} GB/s")

import numpy as np
for N in (1000, 118287):
    corpus = torch.randn(N, 512, device=dev); qs = torch.randn(16, 512, device=dev)
    for Q in (1, 16):
        ms = bench(lambda: ops.cosine_topk(corpus, qs[:Q], 100), iters=20)
        print(f"topk N={N} Q={Q}: {ms*1e3:.1f} us  scan-bytes/s {(N*512*4)/ms/1e6:.0f} GB/s (whole call incl. select)")
