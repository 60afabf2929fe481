#!/usr/bin/env python3
"""A few launches of one GEMM shape with a forced kernel variant, for rocprofv3 --pmc passes.
Usage: pmc_gemm.py M N K variant [a_mode b_mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
M, N, Kd, v = (int(x) for x in sys.argv[1:5])
am, bm = (int(x) for x in sys.argv[5:7]) if len(sys.argv) > 6 else (0, 0)
K.TUNE["enabled"] = False
K.TUNE["force"] = v
dev = "cuda:0"
a = torch.randn((M, Kd) if am == 0 else (Kd, M), device=dev).bfloat16()
b = torch.randn((N, Kd) if bm == 0 else (Kd, N), device=dev).bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    K.gemm(a, b, M, N, Kd, a_mode=am, b_mode=bm, out=o)
torch.cuda.synchronize()
