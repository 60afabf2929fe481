#!/usr/bin/env python3
"""The vendor BLAS (torch.mm -> hipBLASLt) on the clock probe's problem, random against zero operands: how much of the data dependence is
the chip's and how much the kernel's.  usage: vendor_power.py"""
import torch
dev = "cuda:0"
for (M, N, K) in ((4096, 4096, 8192), (8192, 8192, 8192)):
    for kind in ("randn", "zeros"):
        a = (torch.randn(M, K, device=dev) if kind == "randn" else torch.zeros(M, K, device=dev)).bfloat16()
        b = (torch.randn(K, N, device=dev) if kind == "randn" else torch.zeros(K, N, device=dev)).bfloat16()
        for _ in range(5):
            torch.mm(a, b)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                torch.mm(a, b)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        print(f"torch.mm bf16 {M} x {N} x {K} {kind}: {best * 1e3:.1f} us = {2.0 * M * N * K / best / 1e9:.0f} TF/s", flush=True)
