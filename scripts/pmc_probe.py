import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
dev = torch.device("cuda:0")
B, S, H = 8, 5337, 24
D = H * 128
qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
vt = torch.empty(B, H, 128, (S + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
wq = torch.ones(128, device=dev).bfloat16()
cos = torch.ones(S, 64, device=dev); sin = torch.zeros(S, 64, device=dev)
o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
ops.qk_norm_rope_vt(qkv, vt, wq, wq, wq, wq, cos, sin, B, S, H, 3 * D, 1241)
A = torch.randn(42696, 3072, device=dev).bfloat16(); W = (torch.randn(9216, 3072, device=dev) * 0.02).bfloat16()
C = torch.empty(42696, 9216, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.attention(qkv, qkv.view(-1)[D:], vt, o, B, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))
    ops.gemm(A, W, out=C)
torch.cuda.synchronize()
