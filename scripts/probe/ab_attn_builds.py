"""the DiT's attention call (fused q preparation, B = 8, S = 5337, 24 heads) under TWO builds of the library, alternating processes on one box:
    python scripts/probe/ab_attn_builds.py <libA.so> <libB.so>
Each child process times 5 x 10 launches and prints the median; 4 alternations.  Also prints max |a - b| / max |b| of the two outputs."""
import math, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 2 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from domain_rag_amd import ops
    dev = torch.device("cuda:0")
    B, S, H, s_txt = 8, 5337, 24, 1241
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, S, 3 * D, device=dev, generator=g).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(B, H, 128, s_pad, device=dev, dtype=torch.bfloat16)
    w = [(1 + 0.1 * torch.randn(128, device=dev, generator=g)).bfloat16() for _ in range(4)]
    ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    ops.k_norm_rope_vt(qkv, vt, w[1], w[3], cos, sin, B, S, H, 3 * D, s_txt)
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    run = lambda: ops.attention_qprep(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128), w[0], w[2], cos, sin, s_txt)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    torch.save(o.cpu(), os.environ["OUT_PT"])
    print(statistics.median(ts))
    sys.exit(0)
libs = sys.argv[1:3]
res = {l: [] for l in libs}
for rnd in range(4):
    for i, l in enumerate(libs):
        env = dict(os.environ, DRAG_LIB=os.path.abspath(l), OUT_PT=f"/tmp/ab_attn_{i}.pt")
        if os.environ.get(f"ENV{i}"):            # extra environment of build i, e.g. ENV1="DRAG_ATTN_Q64=1"
            env.update(kv.split("=", 1) for kv in os.environ[f"ENV{i}"].split(","))
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        try:
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(out.stdout[-2000:], out.stderr[-2000:]); raise
import torch
a, b = torch.load("/tmp/ab_attn_0.pt").double(), torch.load("/tmp/ab_attn_1.pt").double()
fl = 4.0 * 5337 * 5337 * 128 * 24 * 8
for l in libs:
    m = statistics.median(res[l])
    print(f"{os.path.basename(l)}: {[round(x * 1e3) for x in res[l]]} us  median {m * 1e3:.0f} us = {fl / m / 1e9:.0f} TFLOP/s")
print(f"outputs: max |a - b| / max |b| = {((a - b).abs().max() / b.abs().max()).item():.3e}, equal elements {(a == b).double().mean().item():.4f}")
