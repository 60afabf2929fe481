"""How fast are the 256x256 / 256x128 kernels on shapes that fill the chip exactly (one or two 256x256 tiles per CU)?  This is the
rate a stream-K schedule could approach on the unbalanced BASELINE shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
K.TUNE["enabled"] = False
dev = "cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
for (M, N, Kd) in ((8192, 2048, 768), (8192, 4096, 768), (8192, 2048, 3072), (8192, 4096, 3072), (8256, 3072, 768), (8256, 768, 3072), (8192, 8192, 768)):
    x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = []
    for v in [int(t) for t in os.environ.get("VARS", "1,3,6,7,10").split(",")]:
        K.TUNE["force"] = v
        t = bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out))
        row.append(f"v{v} {t * 1e6:6.1f} us {2.0 * M * N * Kd / t / 1e12:5.0f} TF")
    K.TUNE["force"] = None
    tm = bench(lambda: torch.mm(x, W.t(), out=out))
    print(f"{M}x{N}x{Kd}: " + " | ".join(row) + f" | torch.mm {tm * 1e6:6.1f} us {2.0 * M * N * Kd / tm / 1e12:5.0f} TF")
