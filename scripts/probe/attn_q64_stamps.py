"""round 5 (experiment build, DRAG_LIB=...): shader-clock stamps of workgroup 0 / wave 0 of attention_q64_kernel: how long the wave waits at each tile's
vmcnt(0) + barrier, what a tile costs between barriers, and the block's prologue / epilogue"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops, _lib
dev = torch.device("cuda:0")
B, S, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 5337, 24)))
import math
D = H * 128
qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
s_pad = (S + 63) // 64 * 64
vt = torch.empty(B, H, 128, s_pad, dtype=torch.bfloat16, device=dev)
ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, B, S, H, 3 * D, 0)
out = torch.empty(B, S, D, dtype=torch.bfloat16, device=dev)
ops.set_option("attn_walk", int(os.environ.get("WALK", "0")))
for _ in range(3): ops.attention(qkv, qkv.view(-1)[D:], vt, out, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
lib.drag_debug_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.drag_debug_attn_stamps(buf, 512) == 0
nkv = (S + 63) // 64
st = list(buf)
reach = [st[2 + 2 * i] for i in range(nkv)]; past = [st[3 + 2 * i] for i in range(nkv)]
waits = [past[i] - reach[i] for i in range(nkv)]
work = [reach[i + 1] - past[i] for i in range(nkv - 1)]
print(f"B={B} S={S} H={H}: {nkv} KV tiles (the LAST item of workgroup 0); item top -> loop {st[1] - st[4 + 2 * nkv]} cycles; loop {st[2 + 2 * nkv] - st[1]}; loop end -> output stored {st[3 + 2 * nkv] - st[2 + 2 * nkv]}; the previous item's q fragments were ready {st[4 + 2 * nkv] - st[5 + 2 * nkv]} cycles before this item's top; kernel start -> this item's top {st[4 + 2 * nkv] - st[0]}")
print(f"  per tile: wait at vmcnt(0) + barrier: median {statistics.median(waits[2:]):.0f} (min {min(waits[2:])}, max {max(waits[2:])}); between barriers: median {statistics.median(work[2:]):.0f} (min {min(work[2:])}, max {max(work[2:])})")
print("  first 12 tiles (wait | work):", " ".join(f"{waits[i]}|{work[i]}" for i in range(12)))
print(f"  sum of waits {sum(waits)} = {100 * sum(waits) / (st[3 + 2 * nkv] - st[4 + 2 * nkv]):.1f} % of the block; 64 MFMAs of a tile are 2048 cycles")
