#!/usr/bin/env python3
"""256 x 256 8-phase kernel (variant 11) against the tuned in-tree choice and the vendor BLAS (torch.mm) on the BASELINE shapes.
Random bf16 operands (cdna_hip_programming.md rule 25), bias -> bf16 epilogue, interleaved rounds, median / min over rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
KS = L.MRG_OP_KSTRIDED


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def case(tag, mode, M, N, Kd, splits=(1,), iters=20, rounds=5):
    torch.manual_seed(0)
    if mode == "fwd":
        A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
        kw = dict(bias=torch.randn(N, device=dev))
        ref = lambda: torch.mm(A, B.t())
    elif mode == "dgrad":
        A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(Kd, N, device=dev) * 0.05).bfloat16()
        kw = dict(b_mode=KS)
        ref = lambda: torch.mm(A, B)
    else:
        A = torch.randn(Kd, M, device=dev).bfloat16(); B = torch.randn(Kd, N, device=dev).bfloat16()
        kw = dict(a_mode=KS, b_mode=KS, out_dtype=torch.float32)
        ref = lambda: torch.mm(A.t(), B)
    out = torch.empty(M, N, dtype=kw.pop("out_dtype", torch.bfloat16), device=dev)
    fns = {"vendor": ref}
    K.TUNE["force"] = None
    fns["tuned"] = lambda: K.gemm(A, B, M, N, Kd, out=out, tune_split=(mode == "wgrad"), splitk=K.pick_splitk(M, N, Kd) if mode == "wgrad" else 1, **kw)
    for sk in splits:
        def f(sk=sk):
            K.TUNE["force"] = L.MRG_GEMM_P8
            K.TUNE["enabled"] = False
            try:
                K.gemm(A, B, M, N, Kd, out=out, splitk=sk, **kw)
            finally:
                K.TUNE["force"] = None
                K.TUNE["enabled"] = True
        fns[f"p8/sk{sk}"] = f
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(timed(f, iters))
    fl = 2.0 * M * N * Kd
    line = f"{tag:28s} {mode:5s} {M:6d}x{N:5d}x{Kd:6d} "
    for k, v in res.items():
        v = sorted(v)
        line += f"| {k} {v[len(v) // 2]:7.1f} us {fl / v[len(v) // 2] / 1e6:6.0f} TF "
    print(line, flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "sq"):
        case("square", "fwd", 4096, 4096, 4096)
        case("square", "fwd", 8192, 8192, 8192, iters=5)
    if which in ("all", "c2"):
        for nm, N, Kd in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
            case("c2 " + nm, "fwd", 8256, N, Kd, splits=(1, 2) if Kd > 768 else (1,))
            case("c2 " + nm, "dgrad", 8256, Kd, N, splits=(1, 2) if N > 768 else (1,))
            case("c2 " + nm, "wgrad", N, Kd, 8256, splits=(4, 7, 9))
    if which in ("all", "c4"):
        for nm, N, Kd in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
            case("c4 " + nm, "fwd", 6336, N, Kd, splits=(1, 2) if Kd > 1024 else (1,))
            case("c4 " + nm, "dgrad", 6336, Kd, N, splits=(1, 2) if N > 1024 else (1,))
            case("c4 " + nm, "wgrad", N, Kd, 6336, splits=(3, 4, 6))
    if which in ("all", "c5"):
        for nm, N, Kd in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
            case("c5 " + nm, "fwd", 65792, N, Kd, iters=5)
