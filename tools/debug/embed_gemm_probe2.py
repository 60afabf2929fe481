#!/usr/bin/env python3
"""Where the 30 us of the patch-projection GEMM go: the same product with pieces of its epilogue / row handling removed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = torch.device("cuda", 0)
torch.manual_seed(0)
K.TUNE["enabled"] = False
B, Kk, D, Kp, ntok = 64, 128, 768, 1024, 256
cap, cnt = B * Kk, B * Kk // 2
A = torch.randn(cap, Kp, device=dev).bfloat16()
W = torch.randn(D, Kp, device=dev).bfloat16() * 0.02
bias = torch.randn(D, device=dev)
x0 = torch.empty(B, Kk + 1, D, device=dev)
x16 = torch.empty(cap, D, device=dev, dtype=torch.bfloat16)
x32 = torch.empty(cap, D, device=dev)
rows = torch.randperm(B * (Kk + 1), device=dev)[:cap].int().contiguous()
rows_sorted = torch.sort(rows[:cnt])[0].contiguous()
rows2 = torch.cat([rows_sorted, rows[cnt:]]).contiguous()
toks = torch.randint(0, ntok, (cap,), device=dev).int()
tab = torch.randn(ntok, D, device=dev)
m_dev = torch.tensor([cnt], dtype=torch.int32, device=dev)
def t(name, f):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(16):
            f()
    best = 1e9
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); g.replay(); en.record(); en.synchronize()
        best = min(best, st.elapsed_time(en) / 16 * 1e3)
    print(f"{name:60s} {best:6.1f} us", flush=True)
for v, vn in ((None, "auto"), (L.MRG_GEMM_GLDS64, "glds64"), (L.MRG_GEMM_WS256, "ws256")):
    K.TUNE["force"] = v
    t(f"{vn}: production (m_dev, row_map, add_tab, bias, fp32)", lambda: K.gemm(A, W, cap, D, Kp, lda=Kp, ldb=Kp, out=x0, ldc=D, bias=bias, row_map=rows, add_tab=tab, add_row=toks, m_dev=m_dev))
    t(f"{vn}: sorted output rows", lambda: K.gemm(A, W, cap, D, Kp, lda=Kp, ldb=Kp, out=x0, ldc=D, bias=bias, row_map=rows2, add_tab=tab, add_row=toks, m_dev=m_dev))
    t(f"{vn}: M = {cnt} host-side, same epilogue", lambda: K.gemm(A, W, cnt, D, Kp, lda=Kp, ldb=Kp, out=x0, ldc=D, bias=bias, row_map=rows, add_tab=tab, add_row=toks))
    t(f"{vn}: m_dev, bias -> fp32 rows in place (no map, no table)", lambda: K.gemm(A, W, cap, D, Kp, lda=Kp, ldb=Kp, out=x32, ldc=D, bias=bias, m_dev=m_dev))
    t(f"{vn}: M = {cnt}, bias -> fp32", lambda: K.gemm(A, W, cnt, D, Kp, lda=Kp, ldb=Kp, out=x32, ldc=D, bias=bias))
    t(f"{vn}: M = {cnt}, bias -> bf16", lambda: K.gemm(A, W, cnt, D, Kp, lda=Kp, ldb=Kp, out=x16, ldc=D, bias=bias))
    t(f"{vn}: M = {cap}, bias -> bf16", lambda: K.gemm(A, W, cap, D, Kp, lda=Kp, ldb=Kp, out=x16, ldc=D, bias=bias))
K.TUNE["force"] = None
