"""how many host threads torch's CPU linear wants at the oracle's shapes on this box (the GPU suite is bound by its CPU oracles)"""
import time
import torch
import torch.nn.functional as F
torch.manual_seed(0)
for S in (304, 1124, 5337):
    x = torch.randn(S, 3072).bfloat16(); w = torch.randn(12288, 3072).bfloat16()
    xf, wf = x.float(), w.float()
    q = torch.randn(1, 24, S, 128)
    for nt in (16, 32, 48, 64, 96, 128):
        torch.set_num_threads(nt)
        out = []
        for name, a, b in (("bf16", x, w), ("f32", xf, wf)):
            for _ in range(2):
                F.linear(a, b)
            t = time.time()
            for _ in range(5):
                F.linear(a, b)
            out.append(f"linear {name} {(time.time() - t) / 5 * 1e3:7.1f} ms")
        for name, qq in (("bf16", q.bfloat16()), ("f32", q)):
            F.scaled_dot_product_attention(qq, qq, qq)
            t = time.time()
            for _ in range(3):
                F.scaled_dot_product_attention(qq, qq, qq)
            out.append(f"sdpa {name} {(time.time() - t) / 3 * 1e3:7.1f} ms")
        print(f"S {S:5d} threads {nt:3d}: " + " | ".join(out), flush=True)
# the VAE oracle's 3 x 3 convolutions and GroupNorm
for (C, R) in ((512, 128), (256, 512), (128, 1024)):
    x = torch.randn(1, C, R, R); w = torch.randn(C, C, 3, 3) * 0.02
    for nt in (16, 32, 64, 128):
        torch.set_num_threads(nt)
        out = []
        for name, a, b in (("bf16", x.bfloat16(), w.bfloat16()), ("f32", x, w)):
            F.conv2d(a, b, padding=1)
            t = time.time()
            for _ in range(2):
                F.conv2d(a, b, padding=1)
            out.append(f"conv3x3 {name} {(time.time() - t) / 2 * 1e3:8.1f} ms")
            t = time.time()
            F.group_norm(a, 32)
            out.append(f"groupnorm {name} {(time.time() - t) * 1e3:7.1f} ms")
        print(f"C {C} {R}x{R} threads {nt:3d}: " + " | ".join(out), flush=True)
