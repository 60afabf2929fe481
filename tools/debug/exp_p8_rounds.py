"""What would a tile shape that fills the CUs exactly buy on the K = 768 forward shapes?  The 8-phase 256 x 256 kernel on problems with
exactly 256 / 512 tiles and the same K, against the real shapes (297 / 396 tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = "cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
K.TUNE["enabled"] = False
for (M, N, Kd, tag) in ((8192, 2048, 768, "256 tiles"), (8256, 2304, 768, "qkv 297 tiles"), (8192, 4096, 768, "512 tiles"), (8256, 3072, 768, "fc1 396 tiles"),
                        (8192, 1024, 3072, "128 tiles K=3072"), (8192, 2048, 3072, "256 tiles K=3072"), (8256, 768, 3072, "fc2 99 tiles"),
                        (7936, 2304, 768, "31x9 = 279"), (7168, 2304, 768, "28x9 = 252")):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); b = torch.zeros(N, device=dev)
    for v, name in ((L.MRG_GEMM_P8, "p8"), (L.MRG_GEMM_GLDS3, "glds3"), (L.MRG_GEMM_GLDS4, "glds4"), (L.MRG_GEMM_WS256, "ws256")):
        K.TUNE["force"] = v
        t = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out))
        print(f"{tag:22s} {M}x{N}x{Kd} {name:6s} {t * 1e6:7.1f} us {2.0 * M * N * Kd / t / 1e12:6.0f} TF/s")
