import os, sys
sys.path.insert(0, os.getcwd())
import torch
from mirage_amd import kernels as K, _lib as L
dev="cuda:0"
def graph_time(fn, inner=10, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * inner)
M = 8256
for (N, Kd, gelu) in ((3072, 768, True), (2304, 768, False), (3072, 768, False)):
    A = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(M, N, dtype=torch.float16, device=dev)
    for v in (1, 2):
        K.TUNE["force"] = v; K.TUNE["enabled"] = False
        if gelu: f = lambda: K.gemm(A, W, M, N, Kd, out=out, bias=b, act=L.MRG_ACT_GELU_SAVE_GRAD, aux_out=aux)
        else: f = lambda: K.gemm(A, W, M, N, Kd, out=out, bias=b)
        print(f"{M}x{N}x{Kd} gelu={gelu} variant {v}: {graph_time(f):6.1f} us", flush=True)
K.TUNE["force"] = None
