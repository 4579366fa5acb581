import sys; sys.path.insert(0,'/root/repo')
import torch
from domain_rag_amd import ops
dev=torch.device('cuda:0')
M,N,K=256,512,8192
g=torch.Generator().manual_seed(1)
a=torch.randn(M,K,generator=g).bfloat16().to(dev); w=(torch.randn(N,K,generator=g)*0.05).bfloat16().to(dev); b=torch.randn(N,generator=g).bfloat16().to(dev)
resid=torch.randn(M,N,generator=g).bfloat16().to(dev)
ops.gemm(a[:256],w[:256])
exp=(resid.float()+((a.float()@w.float().T)+b.float()).bfloat16().float()).bfloat16()
for v in (1,8,2):
    ops.set_option("gemm_splitk",v)
    x=resid.clone(); ops.gemm(a,w,out=x,bias=b,resid=x)
    y=torch.empty_like(resid); ops.gemm(a,w,out=y,bias=b,resid=resid)
    print(v,'in place vs expected max diff',(x.float()-exp.float()).abs().max().item(),'| separate out',(y.float()-exp.float()).abs().max().item())
ops.set_option("gemm_splitk",0)
