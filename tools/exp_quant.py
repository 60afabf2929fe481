"""Wave-quantisation experiment: same N, K, different numbers of 128x128 tiles per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
K.TUNE["enabled"] = False
v = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_orig = K._gemm_launch
K._gemm_launch = lambda d, splitk, variant, dev_, defer_out=None: _orig(d, splitk, v, dev_, defer_out)
dev = "cuda:0"
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
for N, Kd in ((768, 3072), (768, 768), (3072, 768), (2304, 768)):
    for M in (2688, 5376, 5504, 8192, 8256, 10752, 10880, 16384):
        x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); b = torch.zeros(N, device=dev)
        t = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out))
        tiles = ((M + 127) // 128) * (N // 128)
        print(f"N={N} K={Kd} M={M:6d} tiles={tiles:5d} ({tiles / 256:5.2f}/CU) {t * 1e6:7.1f} us {2.0 * M * N * Kd / t / 1e12:6.0f} TF/s")
