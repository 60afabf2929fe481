import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L
K.TUNE["enabled"] = False
dev = "cuda:0"
v = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (M, N, Kd) in ((256, 128, 64), (512, 256, 128), (1000, 328, 192), (8192, 2048, 768)):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    K.TUNE["force"] = v
    out = K.gemm(x, W, M, N, Kd, out_dtype=torch.float32)
    torch.cuda.synchronize()
    ref = x.float() @ W.float().t()
    print(M, N, Kd, "max err", float((out - ref).abs().max()), "ref max", float(ref.abs().max()), flush=True)
for (M, N, Kd) in ((256, 128, 64), (512, 256, 128), (8192, 2048, 768)):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for rep in range(3):
        K.gemm(x, W, M, N, Kd, bias=b, out=out)
        torch.cuda.synchronize()
    ref = x.float() @ W.float().t() + b
    print("bf16+bias", M, N, Kd, "max err", float((out.float() - ref).abs().max()), flush=True)
M, N, Kd = 8192, 2048, 768
x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for n in (2, 4, 8, 23):
    for _ in range(n): K.gemm(x, W, M, N, Kd, bias=b, out=out)
    torch.cuda.synchronize()
    print("back-to-back", n, "ok", flush=True)
