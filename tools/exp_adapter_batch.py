#!/usr/bin/env python3
"""Would the output adapters gain from grouped launches?  Time each adapter-shaped kernel at 1x (one adapter: 16384 query rows), 2x
and 3x rows (adapters stacked along the row axis; same weights: an upper bound for a per-group weight pointer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
KS = L.MRG_OP_KSTRIDED


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def gemm_case(tag, M, N, Kd, mode="fwd", gelu=False, resid=False):
    out = []
    for mult in (1, 2, 3):
        Mm = M * mult
        A = torch.randn(Mm, Kd, device=dev).bfloat16()
        if mode == "fwd":
            B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16(); kw = dict(bias=torch.randn(N, device=dev))
        else:
            B = (torch.randn(Kd, N, device=dev) * 0.05).bfloat16(); kw = dict(b_mode=KS)
        if gelu:
            o = torch.empty(Mm, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(Mm, N, dtype=torch.float16, device=dev)
            f = lambda: K.gemm(A, B, Mm, N, Kd, out=o, act=L.MRG_ACT_GELU_SAVE_GRAD, aux_out=aux, **kw)
        elif resid:
            o = torch.empty(Mm, N, dtype=torch.float32, device=dev); r = torch.randn(Mm, N, device=dev)
            f = lambda: K.gemm(A, B, Mm, N, Kd, out=o, resid=r, **kw)
        else:
            o = torch.empty(Mm, N, dtype=torch.bfloat16, device=dev)
            f = lambda: K.gemm(A, B, Mm, N, Kd, out=o, **kw)
        out.append(timed(f))
    print(f"{tag:22s} {mode:5s} {M:6d}x{N:5d}x{Kd:5d}  1x {out[0]:6.1f} us  2x {out[1]:6.1f} us ({out[1] / out[0]:4.2f})  3x {out[2]:6.1f} us ({out[2] / out[0]:4.2f})", flush=True)


def ln_case(M, D):
    out = []
    for mult in (1, 2, 3):
        x = torch.randn(M * mult, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
        out.append(timed(lambda: K.ln_fwd(x, g, b, 1e-6)))
    print(f"ln_fwd {M}x{D}: 1x {out[0]:6.1f}  2x {out[1]:6.1f} ({out[1] / out[0]:4.2f})  3x {out[2]:6.1f} ({out[2] / out[0]:4.2f})", flush=True)


def attn_case(B, H, Nq, Nk, hd):
    out = []
    D = H * hd
    for mult in (1, 2, 3):
        Bm = B * mult
        if Nq == Nk:
            qkv = torch.randn(Bm, Nq, 3 * D, device=dev).bfloat16()
            st = (Nq * 3 * D, 3 * D)
            f = lambda: K.attn_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], Bm, H, Nq, Nk, hd, hd ** -0.5, st, st, st, need_lse=True)
        else:
            q = torch.randn(Bm, Nq, D, device=dev).bfloat16(); kv = torch.randn(Bm, Nk, 2 * D, device=dev).bfloat16()
            f = lambda: K.attn_fwd(q, kv[..., :D], kv[..., D:], Bm, H, Nq, Nk, hd, hd ** -0.5, (Nq * D, D), (Nk * 2 * D, 2 * D), (Nk * 2 * D, 2 * D), need_lse=True)
        out.append(timed(f))
    print(f"attn_fwd B{B} H{H} {Nq}x{Nk} hd{hd}: 1x {out[0]:6.1f}  2x {out[1]:6.1f} ({out[1] / out[0]:4.2f})  3x {out[2]:6.1f} ({out[2] / out[0]:4.2f})", flush=True)


if __name__ == "__main__":
    M = 16384
    gemm_case("q / proj", M, 256, 256)
    gemm_case("proj + resid", M, 256, 256, resid=True)
    gemm_case("qkv (self)", M, 768, 256)
    gemm_case("fc1 gelu", M, 1024, 256, gelu=True)
    gemm_case("fc2 + resid", M, 256, 1024, resid=True)
    gemm_case("out_proj", M, 1024, 256)
    gemm_case("kv (context)", 8256, 512, 256)
    gemm_case("proj_context", 8256, 256, 768)
    gemm_case("dgrad fc2", M, 1024, 256, mode="dgrad")
    gemm_case("dgrad fc1", M, 256, 1024, mode="dgrad")
    gemm_case("dgrad proj", M, 256, 256, mode="dgrad")
    gemm_case("dgrad qkv", M, 256, 768, mode="dgrad")
    ln_case(M, 256)
    attn_case(64, 8, 256, 256, 32)
    attn_case(64, 8, 256, 129, 32)
