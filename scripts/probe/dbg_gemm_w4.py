import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from domain_rag_amd import ops
print("loaded", flush=True)
dev = torch.device("cuda:0")
M, N, K = 512, 512, 256
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
ops.gemm(A, W, out=C); torch.cuda.synchronize(); ref = C.clone()
print("ref ok", flush=True)
ops.set_option("gemm_kernel", int(os.environ.get("W4", "400")))
C.zero_()
ops.gemm(A, W, out=C); torch.cuda.synchronize()
print("w4 ran; equal:", torch.equal(C, ref), (C.float() - ref.float()).abs().max().item(), flush=True)
for K in (512, 3072):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    ops.set_option("gemm_kernel", 0); ops.gemm(A, W, out=C); ref = C.clone()
    ops.set_option("gemm_kernel", int(os.environ.get("W4", "400"))); ops.gemm(A, W, out=C); torch.cuda.synchronize()
    print(K, "equal:", torch.equal(C, ref), (C.float() - ref.float()).abs().max().item(), flush=True)
