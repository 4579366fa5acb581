"""round 5 (experiment build, DRAG_LIB=...): shader-clock stamps of workgroup 0 / wave 0 of gemm_bf16_w4p around the pieces of its tiles:
0 loop top | 1 next tile's state computed | 2 this tile's K-steps 0, 1 landed (vmcnt) | 3 K loop done | 4 epilogue issued"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops, _lib
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 3072, 3072)))
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
ops.set_option("gemm_kernel", 3)
for _ in range(5): ops.gemm(A, W, out=C)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
lib.drag_debug_w4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.drag_debug_w4_stamps(buf, 512) == 0
ntiles = min(64, ((M + 255) // 256 * ((N + 255) // 256) + 255) // 256)
print(f"M={M} N={N} K={K}: {ntiles} tiles of workgroup 0 (cycles of the shader clock; K loop = {K // 64} K-steps)")
names = ["state", "wait", "K loop", "epilogue"]
for t in range(ntiles):
    s = [buf[t * 8 + i] for i in range(5)]
    nxt = buf[(t + 1) * 8] if t + 1 < ntiles else None
    d = [s[i + 1] - s[i] for i in range(4)]
    print(f"  tile {t}: " + " | ".join(f"{n} {v}" for n, v in zip(names, d)) + (f" | to next top {nxt - s[4]}" if nxt else "") + f"   K loop per K-step {d[2] / (K // 64):.0f}")
