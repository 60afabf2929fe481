#!/usr/bin/env python3
"""The patch-projection GEMM of the step's head (engine.embed_fwd: gathered rows, device-side row count, bias + positional row epilogue
at a mapped output row) under every kernel variant that accepts it: us per launch at C2 / C4 sizes, rows = 1/ndom of the capacity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = torch.device("cuda", 0)
torch.manual_seed(0)

def case(tag, B, Kk, ndom, D, Kp, ntok):
    cap = B * Kk
    cnt = cap // ndom
    A = torch.randn(cap, Kp, device=dev).bfloat16()
    W = torch.randn(D, Kp, device=dev).bfloat16() * 0.02
    bias = torch.randn(D, device=dev)
    x0 = torch.empty(B, Kk + 1, D, device=dev)
    rows = torch.randperm(B * (Kk + 1), device=dev)[:cap].int().contiguous()
    toks = torch.randint(0, ntok, (cap,), device=dev).int()
    tab = torch.randn(ntok, D, device=dev)
    m_dev = torch.tensor([cnt], dtype=torch.int32, device=dev)
    res = []
    for name, v, sk in (("auto", None, 1), ("glds3", L.MRG_GEMM_GLDS3, 1), ("glds4", L.MRG_GEMM_GLDS4, 1), ("pipe128", L.MRG_GEMM_PIPE128, 1),
                        ("glds64", L.MRG_GEMM_GLDS64, 1), ("pipe256", L.MRG_GEMM_PIPE256, 1), ("ws256", L.MRG_GEMM_WS256, 1), ("g256", L.MRG_GEMM_G256, 1),
                        ("auto/2", None, 2), ("auto/4", None, 4), ("glds64/2", L.MRG_GEMM_GLDS64, 2), ("pipe128/2", L.MRG_GEMM_PIPE128, 2), ("pipe128/4", L.MRG_GEMM_PIPE128, 4)):
        K.TUNE["force"] = v
        try:
            f = lambda: K.gemm(A, W, cap, D, Kp, lda=Kp, ldb=Kp, out=x0, ldc=D, bias=bias, row_map=rows, add_tab=tab, add_row=toks, m_dev=m_dev, splitk=sk)
            f(); torch.cuda.synchronize()
            if name == "auto":
                ref = x0.clone()
            else:
                err = float((x0 - ref).abs().max())
                if err > 1e-3:
                    name += f"(ERR {err:.2e})"
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(16):
                    f()
            best = 1e9
            for _ in range(5):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record(); g.replay(); en.record(); en.synchronize()
                best = min(best, st.elapsed_time(en) / 16 * 1e3)
            res.append(f"{name} {best:.1f}")
        except Exception as e:
            res.append(f"{name} -- ({str(e)[:40]})")
        finally:
            K.TUNE["force"] = None
    print(f"{tag}: rows {cnt} of {cap}, {D} x {Kp}: " + " | ".join(res), flush=True)

case("c2", 64, 128, 2, 768, 1024, 256)
case("c4", 64, 98, 3, 1024, 1024, 256)
case("c5", 256, 256, 1, 1024, 1024, 256)
