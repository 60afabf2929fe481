"""Cost of the fused epilogues on the fc1-shaped GEMM (8256 x 3072 x 768) and the fc2-dgrad-shaped one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = "cuda:0"
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
M, N, Kd = 8256, 3072, 768
x = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.02).bfloat16(); b = torch.zeros(N, device=dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); pre = torch.empty_like(out)
print("fc1 fwd  bias only        %.1f us" % bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out)))
print("fc1 fwd  bias+gelu        %.1f us" % bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out, act=L.MRG_ACT_GELU)))
print("fc1 fwd  bias+gelu+aux    %.1f us" % bench(lambda: K.gemm(x, W, M, N, Kd, bias=b, out=out, act=L.MRG_ACT_GELU, aux_out=pre)))
dy = torch.randn(M, Kd, device=dev).bfloat16(); W2 = (torch.randn(Kd, N, device=dev) * 0.02).bfloat16()
dx = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
print("fc2 dgrad plain           %.1f us" % bench(lambda: K.gemm(dy, W2, M, N, Kd, b_mode=L.MRG_OP_KSTRIDED, out=dx)))
print("fc2 dgrad gelu'(aux_in)   %.1f us" % bench(lambda: K.gemm(dy, W2, M, N, Kd, b_mode=L.MRG_OP_KSTRIDED, out=dx, act=L.MRG_ACT_GELU_BWD, aux_in=pre)))
r = torch.randn(M, Kd, device=dev); o32 = torch.empty(M, Kd, device=dev); h = torch.randn(M, N, device=dev).bfloat16(); W3 = (torch.randn(Kd, N, device=dev) * 0.02).bfloat16()
b2 = torch.zeros(Kd, device=dev); o16 = torch.empty(M, Kd, dtype=torch.bfloat16, device=dev)
print("fc2 fwd  bias -> bf16     %.1f us" % bench(lambda: K.gemm(h, W3, M, Kd, N, bias=b2, out=o16)))
print("fc2 fwd  bias+resid fp32  %.1f us" % bench(lambda: K.gemm(h, W3, M, Kd, N, bias=b2, out=o32, resid=r)))
