import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
K.TUNE["enabled"] = False
v = int(os.environ.get("MRG_FORCE_VARIANT", "4"))
_orig = K._gemm_launch
K._gemm_launch = lambda d, splitk, variant, dev_, *a_: _orig(d, splitk, v, dev_, *a_)
pad = int(os.environ.get('MRG_PAD', '0'))     # extra elements per operand row: probes L2-channel effects of power-of-two row strides
dev = "cuda:0"
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
for (M, N, Kd) in ((8256, 768, 3072), (8256, 3072, 768), (4096, 4096, 4096)):
    x = torch.randn(M, Kd + pad, device=dev).bfloat16()[:, :Kd]; W = (torch.randn(N, Kd + pad, device=dev) * 0.02).bfloat16()[:, :Kd]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = bench(lambda: K.gemm(x, W, M, N, Kd, out=out))
    print(f"pad={pad} abl={os.environ.get('MRG_GEMM_ABLATE','0')} {M}x{N}x{Kd}: {t*1e6:8.1f} us  {2.0*M*N*Kd/t/1e12:7.1f} TF/s-equivalent")
