#!/usr/bin/env python3
"""Shader clock during the 8-phase main loop, per workgroup: clock64() cycles over wall_clock64() (100 MHz) ticks of one main loop
(mrg_debug_sk_trace slots 30 / 31 against 0 / 1), for operand fills of different entropy and for grids that fill the chip or not.
One tile per workgroup (tiles == grid: no exchange).  usage: clock_probe.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mirage_amd import kernels as K, _lib as L

Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
KS = len(sys.argv) > 2 and sys.argv[2] == "kstrided"        # both operands [K][rows]: the weight-gradient layout
dev = "cuda:0"
K.TUNE["force"] = L.MRG_GEMM_P8SK
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)


def fill(kind, shape):
    torch.manual_seed(1)
    if kind == "zeros":
        return torch.zeros(shape, device=dev).bfloat16()
    if kind == "const":
        return torch.full(shape, 0.5, device=dev).bfloat16()
    if kind == "sign":
        return (torch.randint(0, 2, shape, device=dev).float() * 2 - 1).bfloat16()
    return torch.randn(shape, device=dev).bfloat16()


print(f"8-phase main loop, K = {Kd} ({Kd // 64} k-tiles), one 256 x 256 tile per workgroup, operands {'[K][rows] (weight-gradient layout)' if KS else 'row-major [rows][K]'}")
print("| tiles = workgroups | operands | us per main loop (median) | TF/s of the launch's main loops | shader clock MHz median (min .. max) |")
print("|---|---|---|---|---|")
for tiles_m, tiles_n in ((16, 16), (8, 16), (8, 8), (2, 4)):
    M, N = tiles_m * 256, tiles_n * 256
    P = tiles_m * tiles_n
    for kind in ("randn", "sign", "const", "zeros"):
        A, B = (fill(kind, (Kd, M)), fill(kind, (Kd, N))) if KS else (fill(kind, (M, Kd)), fill(kind, (N, Kd)))
        kw = dict(a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED) if KS else {}
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        K.SK_GRID = P
        for _ in range(5):
            K.gemm(A, B, M, N, Kd, out=out, **kw)
        torch.cuda.synchronize()
        mhz, us = [], []
        for _ in range(5):
            buf.zero_()
            L.lib.mrg_debug_sk_trace(buf.data_ptr())
            K.gemm(A, B, M, N, Kd, out=out, **kw)
            torch.cuda.synchronize()
            L.lib.mrg_debug_sk_trace(None)
            t = buf.cpu().numpy().reshape(-1, 32)
            u = np.nonzero(t[:, 0])[0]
            wall = (t[u, 1] - t[u, 0]).astype(np.float64)           # 10 ns ticks
            cyc = (t[u, 31] - t[u, 30]).astype(np.float64)
            mhz.append(cyc / wall * 100.0)
            us.append(wall / 100.0)
        mhz, us = np.concatenate(mhz), np.concatenate(us)
        tf = 2.0 * 256 * 256 * Kd * P / (np.median(us) * 1e-6) / 1e12
        print(f"| {P} | {kind} | {np.median(us):.1f} | {tf:.0f} | {np.median(mhz):.0f} ({mhz.min():.0f} .. {mhz.max():.0f}) |", flush=True)
