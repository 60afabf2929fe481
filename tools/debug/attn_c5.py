#!/usr/bin/env python3
"""Attention forward at the C5 shapes (whole batch and one of the two half-batch chains), N = 257 against N = 256: what the odd token
costs there, next to the HBM floor (q, k, v read + o written once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K
dev = "cuda:0"; BF16 = torch.bfloat16
def run(B, H, N, hd):
    Dm = H * hd
    qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16); st = (N * 3 * Dm, 3 * Dm)
    q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
    for _ in range(3): K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st, need_lse=False)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st, need_lse=False)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    mb = 4 * B * N * Dm * 2 / 1e6
    fl = 4.0 * B * H * N * N * hd
    print(f"B={B} H={H} N={N} hd={hd}: {best:.1f} us; {mb:.0f} MB -> HBM floor {mb / 6.3e3 * 1e3 / 1e3:.1f} us at 6.3 TB/s; {fl / best / 1e6:.0f} TF/s", flush=True)
for B in (256, 128):
    for N in (257, 256):
        run(B, 16, N, 64)
