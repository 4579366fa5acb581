"""randomised stress of attention_q64g_kernel against attention_q64_kernel: random (B, H, S with an even KV tile count, s_txt), random walks, random
hot keys and score scales; per case: the generated stream without the fold == the hand-placed kernel's bits (plain and fused q preparation), the
fold finite, launch-to-launch deterministic, walk-invariant and within 1e-2 of the unfolded output.  python scripts/probe/attn_gen_stress.py [cases]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from domain_rag_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
scale = 1 / math.sqrt(128)
rng = np.random.default_rng(2026)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bad = 0


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()


try:
    for ci in range(ncases):
        B = int(rng.integers(1, 4)); H = int(rng.choice([1, 2, 3, 4, 8, 8, 16]))
        tiles = int(rng.choice([16, 18, 20, 22, 26, 34, 66, 84])) if ci % 5 else int(rng.choice([4, 6, 8, 10]))
        S = tiles * 64 - int(rng.integers(0, 64))
        if S <= (tiles - 1) * 64:
            S = (tiles - 1) * 64 + 1
        if S < 1024 and tiles >= 16:
            continue
        s_txt = int(rng.integers(0, min(S, 1300)))
        qprep = bool(rng.integers(0, 2))
        D = H * 128
        g = torch.Generator().manual_seed(1000 + ci)
        amp = float(rng.choice([0.5, 1.0, 3.0, 6.0]))
        qkv = torch.randn(B, S, 3 * D, generator=g) * amp
        for _ in range(int(rng.integers(0, 4))):
            b, h, key, qrow = int(rng.integers(0, B)), int(rng.integers(0, H)), int(rng.integers(0, S)), int(rng.integers(0, S))
            qkv[b, key, D + h * 128: D + (h + 1) * 128] = qkv[b, qrow, h * 128:(h + 1) * 128] * float(rng.choice([5.0, 20.0, 60.0]))
        qkv = qkv.bfloat16().to(dev)
        w = [(1 + 0.1 * torch.randn(128, generator=g)).bfloat16().to(dev) for _ in range(4)]
        ang = torch.rand(S, 64, generator=g) * 6.28
        cos, sin = torch.cos(ang).contiguous().to(dev), torch.sin(ang).contiguous().to(dev)
        vt = torch.empty(B, H, 128, tiles * 64, device=dev, dtype=torch.bfloat16)
        x = qkv.clone()
        if qprep:
            ops.k_norm_rope_vt(x, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
        else:
            ops.qk_norm_rope_vt(x, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)

        def run(gen, walk):
            ops.set_option("attn_gen", gen); ops.set_option("attn_walk", walk); ops.set_option("attn_q64", 1 if S >= 1024 else 0)
            o = torch.full((B, S, D), float("nan"), device=dev, dtype=torch.bfloat16)
            if qprep:
                ops.attention_qprep(x, x.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, scale, w[0], w[2], cos, sin, s_txt)
            else:
                ops.attention(x, x.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, scale)
            return o
        if S < 1024:
            continue            # (the 64-query kernels take S >= 1024 only)
        walks = [2, 0] + ([8] if (B * H) % 8 == 0 else [])
        old = run(1, 2)
        msgs = []
        if not torch.isfinite(old.float()).all():
            msgs.append("hand-placed kernel non-finite")
        for wk in walks:
            got = run(2, wk)
            if not torch.equal(got, old):
                msgs.append(f"no-fold walk {wk}: bits differ (rel {rel(got, old):.2e})")
        if qprep:
            f0 = run(0, 2)
            if not torch.isfinite(f0.float()).all():
                msgs.append("fold non-finite")
            for wk in walks:
                f = run(0, wk)
                if not torch.equal(f, f0):
                    msgs.append(f"fold walk {wk}: differs from walk 2")
            r = rel(f0, old)
            if not r < 1.2e-2:
                msgs.append(f"fold vs unfolded rel {r:.2e}")
        status = "ok" if not msgs else "FAIL " + "; ".join(msgs)
        bad += bool(msgs)
        print(f"case {ci:3d}: B {B} H {H:2d} S {S:5d} ({tiles} tiles) s_txt {s_txt:4d} qprep {int(qprep)} amp {amp}: {status}", flush=True)
finally:
    ops.set_option("attn_gen", 0); ops.set_option("attn_walk", 0); ops.set_option("attn_q64", 0)
print(f"{bad} failing cases" if bad else "ALL OK", flush=True)
