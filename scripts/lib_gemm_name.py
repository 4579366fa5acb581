import torch
dev = torch.device("cuda:0")
for (M, N, K) in ((32768, 3072, 3072), (32768, 9216, 3072), (42696, 3072, 15360), (9928, 3072, 12288)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3): torch.matmul(A, W.t())
    torch.cuda.synchronize()
