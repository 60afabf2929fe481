#!/usr/bin/env python3
"""Micro-benchmarks of the non-GEMM kernels at the C2 shapes (HBM GB/s or TF/s per kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = "cuda:0"
BF16, F32 = torch.bfloat16, torch.float32

def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters

which = sys.argv[1:] or ["ln", "colsum", "attn"]
M, D = 8256, 768
if "ln" in which:
    x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    t = bench(lambda: K.ln_fwd(x, g, b, 1e-6))
    print(f"ln_fwd  {M}x{D}: {t*1e6:7.1f} us  {(M*D*6)/t/1e9:7.0f} GB/s")
    y, mean, rstd = K.ln_fwd(x, g, b, 1e-6)
    dy = torch.randn(M, D, device=dev).to(BF16); dres = torch.randn(M, D, device=dev)
    dg = torch.empty(D, device=dev); db = torch.empty(D, device=dev); dc = torch.empty(D, device=dev)
    t = bench(lambda: K.ln_bwd(dy, x, mean, rstd, g, dres, dg, db, dc))
    print(f"ln_bwd  {M}x{D}: {t*1e6:7.1f} us  {(M*D*16)/t/1e9:7.0f} GB/s (incl. finishing kernel)")
if "colsum" in which:
    for N in (768, 2304, 3072):
        xx = torch.randn(M, N, device=dev).to(BF16); out = torch.empty(N, device=dev)
        t = bench(lambda: K.colsum(xx, out))
        print(f"colsum  {M}x{N}: {t*1e6:7.1f} us  {(M*N*2)/t/1e9:7.0f} GB/s")
if "attn" in which:
    for (B, H, N, hd) in ((64, 12, 129, 64), (64, 16, 99, 64), (64, 8, 256, 32), (256, 16, 257, 64)):
        Dm = H * hd
        qkv = torch.randn(B, N, 3 * Dm, device=dev).to(BF16)
        st = (N * 3 * Dm, 3 * Dm)
        q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
        fl = 4.0 * B * H * N * N * hd
        t = bench(lambda: K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st))
        o, lse = K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st)
        d_o = torch.randn(B, N, Dm, device=dev).to(BF16); dqkv = torch.empty_like(qkv)
        tb = bench(lambda: K.attn_bwd(q, k, v, o, lse, d_o, dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:], B, H, N, N, hd, hd ** -0.5, st, st, st, st, st, st))
        print(f"attn B{B} H{H} N{N} hd{hd}: fwd {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF/s | bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s")
