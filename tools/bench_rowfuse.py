#!/usr/bin/env python3
"""The output adapters' fused row chains (mrg_rowfuse_fwd) against the separate launches they replace, alone on the chip: 20 repetitions
inside a captured graph over rotating operand sets (training side outputs on).  us per chain, HBM bytes per chain / time."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
BF16, F32 = torch.bfloat16, torch.float32


def lin(nout, nin):
    l = types.SimpleNamespace()
    l.w16 = (torch.randn(nout, nin, device=dev) * nin ** -0.5).bfloat16()
    l.bias = torch.randn(nout, device=dev) * 0.1
    return l


def norm():
    n = types.SimpleNamespace()
    n.weight = 1 + 0.1 * torch.randn(256, device=dev); n.bias = 0.1 * torch.randn(256, device=dev); n.eps = 1e-6
    return n


def timed(run, nrot, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(nrot):
            run(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(reps):
                run(i % nrot)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay(); a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[2]


save = os.environ.get("SAVE", "1") != "0"
for M in (16384, 8256, 6336):
    nrot = 3
    proj, n, fc1, fc2 = lin(256, 256), norm(), lin(1024, 256), lin(256, 1024)
    os_ = [torch.randn(M, 256, device=dev).bfloat16() for _ in range(nrot)]
    rs_ = [torch.randn(M, 256, device=dev) for _ in range(nrot)]

    def fused(i):
        K.rowfuse_mlp(os_[i], proj, proj.w16, rs_[i], n, fc1, fc1.w16, fc2, fc2.w16, save, True)

    def separate(i):
        x2 = K.gemm(os_[i], proj.w16, M, 256, 256, bias=proj.bias, resid=rs_[i], out_dtype=F32)
        ln2, _, _ = K.ln_fwd(x2, n.weight, n.bias, n.eps)
        hg = torch.empty((M, 1024), dtype=torch.float16, device=dev) if save else None
        h2 = K.gemm(ln2, fc1.w16, M, 1024, 256, bias=fc1.bias, out_dtype=BF16, act=L.MRG_ACT_GELU_SAVE_GRAD if save else L.MRG_ACT_GELU, aux_out=hg)
        out2 = torch.empty((M, 256), dtype=BF16, device=dev)
        K.gemm(h2, fc2.w16, M, 256, 1024, bias=fc2.bias, resid=x2, out_dtype=F32, out2=out2)

    tf, ts = timed(fused, nrot), timed(separate, nrot)
    flops = 2.0 * M * (256 * 256 + 2 * 256 * 1024)
    by_f = M * (512 + 1024 + 1024 + 512 + (1024 + 512 + 2048 + 2048 if save else 0))     # o, resid | y32, y16 | x, ln, h, h'
    print(f"post-attention chain M={M:6d} save={int(save)}: fused {tf:6.1f} us ({flops / tf / 1e6:5.0f} TF/s, {by_f / tf / 1e6:4.2f} TB/s)   "
          f"separate (4 launches) {ts:6.1f} us   x{ts / tf:4.2f}", flush=True)
    for NH in (256, 512, 768):
        l = lin(NH, 256)
        xs = [torch.randn(M, 256, device=dev) for _ in range(nrot)]

        def fused1(i):
            K.rowfuse_ln_proj(xs[i], n, l, l.w16, save)

        def separate1(i):
            ln2, _, _ = K.ln_fwd(xs[i], n.weight, n.bias, n.eps)
            K.gemm(ln2, l.w16, M, NH, 256, bias=l.bias, out_dtype=BF16)

        tf, ts = timed(fused1, nrot), timed(separate1, nrot)
        print(f"  norm + projection M={M:6d} NH={NH:4d}: fused {tf:6.1f} us   separate (2 launches) {ts:6.1f} us   x{ts / tf:4.2f}", flush=True)
