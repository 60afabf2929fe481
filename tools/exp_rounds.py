#!/usr/bin/env python3
"""Tile-count quantisation experiment: every kernel variant on forward shapes with M = 8192 / 8256 and N chosen so that the 256 x 256
kernel has exactly 1 / 1.125 / 1.5 / 2 rounds of tiles (K = 768 and 3072).  Bias -> bf16 epilogue, random data, hot loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
NAMES = {1: "glds3", 2: "glds4", 3: "pipe128", 4: "pipe256", 7: "ws256", 10: "g256", 11: "p8"}


def timed(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def case(M, N, Kd):
    torch.manual_seed(0)
    A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {}
    fns = {"vendor": lambda: torch.mm(A, B.t())}
    for v, nm in NAMES.items():
        def f(v=v):
            K.TUNE["force"] = v; K.TUNE["enabled"] = False
            try:
                K.gemm(A, B, M, N, Kd, out=out, bias=bias)
            finally:
                K.TUNE["force"] = None; K.TUNE["enabled"] = True
        fns[nm] = f
    for k, f in fns.items():
        try:
            f(); f()
        except Exception as exc:
            res[k] = None
    torch.cuda.synchronize()
    for _ in range(3):
        for k, f in fns.items():
            if res.get(k, 0) is None:
                continue
            res.setdefault(k, []).append(timed(f))
    fl = 2.0 * M * N * Kd
    t8 = ((M + 255) // 256) * ((N + 255) // 256)
    line = f"{M:6d}x{N:5d}x{Kd:5d} tiles256 {t8:4d} ({t8 / 256:4.2f} rounds) "
    for k, v in res.items():
        if v:
            m = sorted(v)[len(v) // 2]
            line += f"| {k} {m:6.1f} us {fl / m / 1e6:5.0f} "
    print(line, flush=True)


if __name__ == "__main__":
    for Kd in (768, 3072):
        for M in (8192, 8256):
            for N in ((768, 1024, 2048, 2304, 3072, 4096) if Kd == 768 else (768, 1024, 2048)):
                case(M, N, Kd)
