#!/usr/bin/env python3
"""Attention forward / backward at the encoder shapes, timed inside a captured graph (20 launches per replay: eager timing of ~20 us
kernels is host-bound): fused-qkv layout vs per-head contiguous layout, N = 129 vs 128 (what the one-token tail tile costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K

dev = "cuda:0"


def graph_time(fn, inner=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * inner)


def case(B, H, N, hd, layout):
    D = H * hd
    if layout == "fused":
        qkv = torch.randn(B, N, 3 * D, device=dev).bfloat16()
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        st = (N * 3 * D, 3 * D)
        o_st = None
    else:   # [B, H, N, hd] contiguous per head: row stride hd, "batch" stride H*N*hd with head offset folded by passing H=1, B=B*H
        q = torch.randn(B * H, N, hd, device=dev).bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q)
        st = (N * hd, hd)
    if layout == "fused":
        f = lambda: K.attn_fwd(q, k, v, B, H, N, N, hd, hd ** -0.5, st, st, st, need_lse=True)
    else:
        f = lambda: K.attn_fwd(q, k, v, B * H, 1, N, N, hd, hd ** -0.5, st, st, st, need_lse=True)
    t_f = graph_time(f)
    o, lse = f()
    d_o = torch.randn_like(o)
    if layout == "fused":
        dqkv = torch.empty_like(qkv)
        fb = lambda: K.attn_bwd(q, k, v, o, lse, d_o, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], B, H, N, N, hd, hd ** -0.5, st, st, st, st, st, st)
    else:
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        fb = lambda: K.attn_bwd(q, k, v, o, lse, d_o, dq, dk, dv, B * H, 1, N, N, hd, hd ** -0.5, st, st, st, st, st, st)
    t_b = graph_time(fb)
    mb_f = B * H * N * hd * 2 * 4 / 1e6
    mb_b = B * H * N * hd * 2 * 8 / 1e6
    print(f"B{B} H{H} N{N} hd{hd} {layout:6s}: fwd {t_f:6.1f} us ({mb_f / t_f * 1e-6 * 1e6 / 1e6:5.2f} TB/s of {mb_f:5.1f} MB)   bwd {t_b:6.1f} us ({mb_b / t_b:5.2f} TB/s of {mb_b:5.1f} MB)", flush=True)


if __name__ == "__main__":
    for N in (129, 128, 160):
        for layout in ("fused", "heads"):
            case(64, 12, N, 64, layout)
    case(32, 12, 129, 64, "fused")
    case(64, 16, 99, 64, "fused")
    case(64, 16, 96, 64, "fused")
    case(64, 8, 256, 32, "fused")
