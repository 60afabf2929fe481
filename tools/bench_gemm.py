#!/usr/bin/env python3
"""Micro-benchmark of mrg_gemm on the shapes of the C2 / C4 training step (forward, dgrad, wgrad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
import os as _os
if _os.environ.get("MRG_FORCE_VARIANT"):
    K.TUNE["enabled"] = False
    _v = int(_os.environ["MRG_FORCE_VARIANT"])
    _orig = K._gemm_launch
    K._gemm_launch = lambda d, splitk, variant, dev_, *a_: _orig(d, splitk, _v, dev_, *a_)
elif _os.environ.get("MRG_NO_TUNE"):
    K.TUNE["enabled"] = False
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters

def run(M, D, tag):
    rows = []
    for name, N, Kd in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16(); b = torch.zeros(N, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(M, Kd, dtype=torch.bfloat16, device=dev)
        dW = torch.empty(N, Kd, device=dev)
        fl = 2.0 * M * N * Kd
        t = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out))
        rows.append((f"{tag} {name} fwd  {M}x{N}x{Kd}", fl / t / 1e12, t * 1e6))
        t = bench(lambda: K.gemm(dy, W, M, Kd, N, b_mode=L.MRG_OP_KSTRIDED, out=dx))
        rows.append((f"{tag} {name} dgrad {M}x{Kd}x{N}", fl / t / 1e12, t * 1e6))
        sk = K.pick_splitk(N, Kd, M)
        t = bench(lambda: K.gemm(dy, x, N, Kd, M, a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED, out=dW, splitk=sk))
        rows.append((f"{tag} {name} wgrad {N}x{Kd}x{M} splitk={sk}", fl / t / 1e12, t * 1e6))
    for r in rows:
        print(f"{r[0]:50s} {r[1]:8.1f} TF/s {r[2]:9.1f} us")
    tot_fl = sum(r[1] * r[2] for r in rows); tot_t = sum(r[2] for r in rows)
    print(f"{tag} block total: {tot_t:.1f} us, {tot_fl / tot_t:.1f} TF/s\n")

if __name__ == "__main__":
    run(8256, 768, "C2-base")
    run(6336, 1024, "C4-large")
    # decoder shapes (Dd = 256, B*256 = 16384 rows)
    run(16384, 256, "decoder")
    # square reference
    for n in (4096,):
        a = torch.randn(n, n, device=dev).bfloat16(); bm = torch.randn(n, n, device=dev).bfloat16(); o = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
        t = bench(lambda: K.gemm(a, bm, n, n, n, out=o))
        print(f"square {n}: {2.0 * n ** 3 / t / 1e12:.1f} TF/s")
