#!/usr/bin/env python3
"""The encoder GEMM shapes of C2 / C4 / C5 WITH their production epilogues (bias -> bf16, bias + fp32 residual, bias + GELU + fp16 GELU' side
output, multiply by the GELU' side input): the tuned table entry, the 8-phase kernel (variant 11) and the vendor BLAS
(plain epilogue).  Every variant is timed as a captured graph of `reps` launches that rotate through `nrot` operand / output sets (cold-ish
operands as inside the step), interleaved rounds, median.  (Round 4 used this tool for the stream-K schedule, hence the name.)

usage: bench_sk.py [c2|c4|c5|all|epi|resid]
  epi    one shape per configuration under every epilogue kind: what the epilogue costs on top of the main loop (MRG_BENCH_ZERO=1: zero operands)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
KS = L.MRG_OP_KSTRIDED
BF16, F32 = torch.bfloat16, torch.float32


def graph_of(fn, reps):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(2):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps):
                fn(i)
    torch.cuda.synchronize()
    return g


def time_graph(g, reps):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    st.record(); g.replay(); en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / reps


def case(tag, mode, epi, M, N, Kd, grids, reps=16, nrot=4, rounds=5):
    torch.manual_seed(0)
    sets = []
    for r in range(nrot):
        if mode == "fwd":
            A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
            kw = {}
        else:
            A = torch.randn(M, Kd, device=dev).bfloat16(); B = (torch.randn(Kd, N, device=dev) * 0.05).bfloat16()
            kw = dict(b_mode=KS)
        if epi == "bias":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=BF16, device=dev))
        elif epi == "plain":
            kw.update(out=torch.empty(M, N, dtype=BF16, device=dev))
        elif epi == "resid":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=F32, device=dev), resid=torch.randn(M, N, device=dev))
        elif epi == "gelu":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=BF16, device=dev), act=L.MRG_ACT_GELU_SAVE_GRAD,
                      aux_out=torch.empty(M, N, dtype=torch.float16, device=dev))
        elif epi == "resid0":      # experiment: the residual is ONE row broadcast (row stride 0: always cache-resident) -- what the residual READ costs
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=F32, device=dev), resid=torch.randn(1, N, device=dev).expand(M, N))
        elif epi == "gelu5":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=BF16, device=dev), act=L.MRG_ACT_GELU)
        elif epi == "f32":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=F32, device=dev))
        elif epi == "mulaux":
            kw.update(out=torch.empty(M, N, dtype=BF16, device=dev), act=L.MRG_ACT_MUL_AUX_F16,
                      aux_in=torch.rand(M, N, device=dev).half())
        sets.append((A, B, kw))

    if os.environ.get("MRG_BENCH_ZERO") == "1":      # zero operands: the matrix units draw less power, the clock stays high (power vs issue experiments)
        for A, B, kw in sets:
            A.zero_(); B.zero_()

    def runner(variant, splitk=1):
        def f(i):
            A, B, kw = sets[i % nrot]
            if splitk > 1:
                kw = dict(kw, splitk=splitk)
            if variant == "vendor":
                torch.mm(A, B.t() if mode == "fwd" else B)
                return
            K.TUNE["force"] = variant
            was = K.TUNE["enabled"]
            if splitk > 1:
                K.TUNE["enabled"] = False          # the table would overwrite the split factor
            try:
                K.gemm(A, B, M, N, Kd, **kw)
            finally:
                K.TUNE["force"] = None
                K.TUNE["enabled"] = was
        return f

    fns = {"vendor": runner("vendor"), "tuned": runner(None), "p8": runner(L.MRG_GEMM_P8)}
    graphs = {k: graph_of(f, reps) for k, f in fns.items()}
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, g in graphs.items():
            res[k].append(time_graph(g, reps))
    fl = 2.0 * M * N * Kd
    line = f"{tag:10s} {mode:5s} {epi:6s} {M:6d}x{N:5d}x{Kd:5d} "
    for k, v in res.items():
        v = sorted(v)
        m = v[len(v) // 2]
        line += f"| {k} {m:6.1f} us {fl / m / 1e6:5.0f} TF "
    print(line, flush=True)


def shapes(M, D):
    H = 4 * D
    return [("qkv", "fwd", "bias", M, 3 * D, D), ("proj", "fwd", "resid", M, D, D), ("fc1", "fwd", "gelu", M, H, D),
            ("fc2", "fwd", "resid", M, D, H), ("fc2", "dgrad", "mulaux", M, H, D), ("fc1", "dgrad", "plain", M, D, H),
            ("proj", "dgrad", "plain", M, D, D), ("qkv", "dgrad", "plain", M, D, 3 * D)]


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    grids = []
    if which == "resid":        # what bounds the fp32-residual epilogue: the residual read (resid0 = cache-resident broadcast row), the fp32 write (f32), or neither
        for tag, M, N, Kd in (("c5h proj", 32896, 1024, 1024), ("c5h fc2", 32896, 1024, 4096), ("c2 fc2", 8256, 768, 3072), ("c4 fc2", 6336, 1024, 4096), ("c4 proj", 6336, 1024, 1024)):
            for epi in ("bias", "f32", "resid0", "resid"):
                case(tag, "fwd", epi, M, N, Kd, grids, reps=(6 if M > 30000 else 16), nrot=(2 if M > 30000 else 4))
    if which == "epi":          # one shape per configuration under every epilogue: what the epilogue costs on top of the main loop
        for tag, M, N, Kd in (("c2 fc1", 8256, 3072, 768), ("c5h fc1", 32896, 4096, 1024)):
            for epi in ("plain", "bias", "gelu5", "gelu", "mulaux", "f32", "resid"):
                case(tag, "fwd", epi, M, N, Kd, grids, reps=(6 if M > 30000 else 16), nrot=(2 if M > 30000 else 4))
    if which in ("all", "c2"):
        for (nm, mode, epi, M, N, Kd) in shapes(8256, 768):
            case("c2 " + nm, mode, epi, M, N, Kd, grids)
        for (nm, mode, epi, M, N, Kd) in shapes(4128, 768)[:4]:
            case("c2h " + nm, mode, epi, M, N, Kd, grids)
    if which in ("all", "c4"):
        for (nm, mode, epi, M, N, Kd) in shapes(6336, 1024):
            case("c4 " + nm, mode, epi, M, N, Kd, grids)
    if which in ("all", "c5"):
        for (nm, mode, epi, M, N, Kd) in shapes(65792 // 2, 1024)[:4]:
            case("c5h " + nm, mode, epi, M, N, Kd, grids, reps=6, nrot=2)
