#!/usr/bin/env python3
"""Launch a few GEMM shapes a handful of times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = "cuda:0"
M, D = 8256, 768
for name, N, Kd in (("fc1", 3072, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16(); b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); dx = torch.empty(M, Kd, dtype=torch.bfloat16, device=dev); dW = torch.empty(N, Kd, device=dev)
    for _ in range(5):
        K.gemm(x, W, M, N, Kd, bias=b, out=out)
        K.gemm(dy, W, M, Kd, N, b_mode=L.MRG_OP_KSTRIDED, out=dx)
        K.gemm(dy, x, N, Kd, M, a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED, out=dW, splitk=2)
torch.cuda.synchronize()
