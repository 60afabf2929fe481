"""Which hipBLASLt kernels torch.mm picks for the calibration shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
dev = "cuda:0"
for (M, N, K) in ((8192, 8192, 8192), (65792, 4096, 1024), (65792, 1024, 4096), (8256, 3072, 768), (8256, 768, 3072)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        C = torch.mm(A, B.t())
    torch.cuda.synchronize()
