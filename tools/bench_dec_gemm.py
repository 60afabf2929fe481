"""Decoder-sized GEMMs (Dd = 256, B*256 = 16384 rows): variant sweep with and without the fused epilogues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
K.TUNE["enabled"] = False
dev = "cuda:0"
_orig = K._gemm_launch
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
for (M, N, Kd) in ((16384, 1024, 256), (16384, 256, 1024), (16384, 256, 256), (16384, 768, 256), (16384, 512, 256), (8256, 768, 768), (8256, 768, 3072), (8256, 2304, 768), (6336, 1024, 1024)):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(M, N, dtype=torch.float16, device=dev)
    row = []
    for v in (1, 2, 3, 7, 8, 10):
        K._gemm_launch = lambda d, splitk, variant, dev_, *a_, v=v: _orig(d, splitk, v, dev_, *a_)
        t0 = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out))
        t1 = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out, act=L.MRG_ACT_GELU_SAVE_GRAD, aux_out=aux))
        row.append(f"v{v}: {t0:5.1f}/{t1:5.1f}")
    mb = (M * Kd + N * Kd + M * N) * 2 / 1e6
    print(f"{M}x{N}x{Kd} ({2.0 * M * N * Kd / 1e9:.1f} GF, {mb:.0f} MB min) us bias/gelu+aux: " + "  ".join(row))
