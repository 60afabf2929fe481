import os, sys
sys.path.insert(0, os.getcwd())
import torch
from mirage_amd import kernels as K
dev="cuda:0"
def timed(fn, iters=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
for M, D in ((16384, 256), (8256, 256), (32768, 256)):
    x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y, mean, rstd = K.ln_fwd(x, g, b, 1e-6)
    dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev)
    dg = torch.empty(D, device=dev); db = torch.empty(D, device=dev); dc = torch.empty(D, device=dev)
    print(M, D, "fwd %.1f us  bwd %.1f us" % (timed(lambda: K.ln_fwd(x, g, b, 1e-6)), timed(lambda: K.ln_bwd(dy, x, mean, rstd, g, dres, dg, db, dc))), flush=True)
