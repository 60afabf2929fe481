#!/usr/bin/env python3
"""LayerNorm backward (mrg_ln_bwd: dx = dres + LN'(dy), bf16 twin, partial dgamma / dbeta / column sums) at the encoder shapes, 20 launches
inside a captured graph over rotating operand sets; GB/s against the 16 B per element the launch moves.  MRG_LIB_PATH selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K

dev = "cuda:0"
for (M, D) in ((8256, 768), (6336, 1024), (4128, 768), (3168, 1024), (65792, 1024)):
    nrot = 3
    sets = []
    for r in range(nrot):
        x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev)
        mean = x.mean(-1); rstd = 1.0 / (x.var(-1, unbiased=False) + 1e-6).sqrt()
        g = torch.randn(D, device=dev)
        sets.append((dy, x, mean.contiguous(), rstd.contiguous(), g, dres, torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)))
    def run(i):
        dy, x, mean, rstd, g, dres, dg, db, dc = sets[i % nrot]
        K.ln_bwd(dy, x, mean, rstd, g, dres, dg, db, dc)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(0); run(1); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(20):
                run(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay(); a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    t = sorted(ts)[2]
    by = M * D * 16.0
    print(f"ln_bwd {M:6d} x {D:4d}: {t:6.1f} us  {by / t / 1e6:6.2f} TB/s", flush=True)
