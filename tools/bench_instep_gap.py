#!/usr/bin/env python3
"""Why the same GEMM kernel is slower inside the step than in a tight loop.

For every encoder GEMM of the C2 step, time the tuned kernel with its REAL epilogue in four settings:
  hot     : one set of buffers, back-to-back launches (what tools/bench_gemm.py / bench_sq.py measure);
  rot_out : the output (and side output) rotates through a pool larger than the 256 MiB Infinity Cache: every launch writes fresh lines;
  rot_all : inputs rotate as well (activations produced "elsewhere", weights of another layer): cold operands + fresh outputs;
  rot_all + a streaming kernel between launches (LayerNorm-like 100 MB pass) that evicts the L2s.
The in-step figure of the same shape comes from `MRG_GEMM_TABLE=1 python bench.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
KS = L.MRG_OP_KSTRIDED
POOL = 12


def run(tag, mk, iters=24):
    """mk(i) -> zero-arg launcher using buffer set i"""
    res = {}
    scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)       # 256 MiB
    for mode in (("hot", "rot_A", "rot_B", "rot_all") if os.environ.get("GAP_AB") else ("hot", "rot_out", "rot_all", "rot_all+evict")):
        fns = [mk(i, mode) for i in range(POOL)]
        for f in fns:
            f()
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            evs = []
            for i in range(iters):
                if mode == "rot_all+evict":
                    scratch.add_(1.0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fns[i % POOL](); e.record()
                evs.append((s, e))
            torch.cuda.synchronize()
            ts.append(sorted(s.elapsed_time(e) * 1e3 for s, e in evs)[len(evs) // 2])
        res[mode] = min(ts)
    print(f"{tag:34s} " + " | ".join(f"{m} {v:6.1f} us" for m, v in res.items()), flush=True)


def main():
    M, D = 8256, 768
    torch.manual_seed(0)
    def bufs(shape, dtype, n=POOL, scale=1.0):
        return [(torch.randn(*shape, device=dev) * scale).to(dtype) for _ in range(n)]
    for name, N, Kd, kind in (("qkv fwd", 3 * D, D, "bias"), ("proj fwd", D, D, "resid"), ("fc1 fwd", 4 * D, D, "gelu"), ("fc2 fwd", D, 4 * D, "resid")):
        A = bufs((M, Kd), torch.bfloat16); W = bufs((N, Kd), torch.bfloat16, scale=0.05); b = torch.randn(N, device=dev)
        out16 = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(POOL)]
        out32 = [torch.empty(M, N, device=dev) for _ in range(POOL)] if kind == "resid" else None
        res = bufs((M, N), torch.float32) if kind == "resid" else None
        aux = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(POOL)] if kind == "gelu" else None
        def mk(i, mode):
            io = 0 if mode in ("hot", "rot_A", "rot_B") else i
            ia = i if (mode.startswith("rot_all") or mode == "rot_A") else 0
            iw = i if (mode.startswith("rot_all") or mode == "rot_B") else 0
            if kind == "bias":
                return lambda: K.gemm(A[ia], W[iw], M, N, Kd, bias=b, out=out16[io])
            if kind == "resid":
                return lambda: K.gemm(A[ia], W[iw], M, N, Kd, bias=b, out=out32[io], resid=res[ia])
            return lambda: K.gemm(A[ia], W[iw], M, N, Kd, bias=b, out=out16[io], act=L.MRG_ACT_GELU_SAVE_GRAD, aux_out=aux[io])
        run(f"{name} {M}x{N}x{Kd} ({kind})", mk)
        del A, W, out16, out32, res, aux
    # dgrad of fc2 (multiply by GELU' + column sums) and of fc1
    for name, N, Kd, kind in (("fc2 dgrad", 4 * D, D, "mulaux"), ("fc1 dgrad", D, 4 * D, "plain"), ("qkv dgrad", D, 3 * D, "plain")):
        dy = bufs((M, Kd), torch.bfloat16); W = bufs((Kd, N), torch.bfloat16, scale=0.05)
        out16 = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(POOL)]
        aux = bufs((M, N), torch.float16) if kind == "mulaux" else None
        cs = torch.empty(N, device=dev)
        def mk(i, mode):
            io = 0 if mode == "hot" else i
            ii = i if mode.startswith("rot_all") else 0
            if kind == "mulaux":
                return lambda: K.gemm(dy[ii], W[ii], M, N, Kd, b_mode=KS, out=out16[io], act=L.MRG_ACT_MUL_AUX_F16, aux_in=aux[ii], colsum=cs)
            return lambda: K.gemm(dy[ii], W[ii], M, N, Kd, b_mode=KS, out=out16[io])
        run(f"{name} {M}x{N}x{Kd} ({kind})", mk)
        del dy, W, out16, aux
    for name, N, Kd in (("fc1 wgrad", 4 * D, D), ("fc2 wgrad", D, 4 * D), ("qkv wgrad", 3 * D, D)):
        dy = bufs((M, N), torch.bfloat16); x = bufs((M, Kd), torch.bfloat16)
        gw = [torch.empty(N, Kd, device=dev) for _ in range(POOL)]
        def mk(i, mode):
            io = 0 if mode == "hot" else i
            ii = i if mode.startswith("rot_all") else 0
            return lambda: K.gemm(dy[ii], x[ii], N, Kd, M, a_mode=KS, b_mode=KS, out=gw[io], splitk=K.pick_splitk(N, Kd, M), tune_split=True)
        run(f"{name} {N}x{Kd}x{M}", mk)
        del dy, x, gw


if __name__ == "__main__":
    main()
