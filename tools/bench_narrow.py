#!/usr/bin/env python3
"""The narrow GEMMs of the output adapters (M = B x 256 query tokens or B x 129 context tokens, N and K in {256, 512, 768, 1024}) with their
production epilogues: every kernel variant forced, the tuned table entry, and the vendor BLAS (plain epilogue), each as a captured graph of
16 launches over rotating operand sets (tools/bench_sk.py's protocol).  These launches run at 110-320 TFLOP/s inside the step
(profiles/r05_table_c2.txt) -- 2-5x their HBM floor -- and add up to 1.3 ms of the 9.2 ms C2 step.
usage: bench_narrow.py [variants ...]     (default: 1 2 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K, _lib as L
import bench_sk as S

variants = [int(v) for v in sys.argv[1:]] or [1, 2, 3]
NAMES = {1: "glds3", 2: "glds4", 3: "pipe128", 7: "ws256", 11: "p8"}


def case(tag, mode, epi, M, N, Kd, reps=16, nrot=4, rounds=5):
    torch.manual_seed(0)
    dev = S.dev
    sets = []
    for r in range(nrot):
        A = torch.randn(M, Kd, device=dev).bfloat16()
        B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16() if mode == "fwd" else (torch.randn(Kd, N, device=dev) * 0.05).bfloat16()
        kw = {} if mode == "fwd" else dict(b_mode=S.KS)
        if epi == "bias":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=S.BF16, device=dev))
        elif epi == "plain":
            kw.update(out=torch.empty(M, N, dtype=S.BF16, device=dev))
        elif epi == "f32":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=S.F32, device=dev))
        elif epi == "plainf32":
            kw.update(out=torch.empty(M, N, dtype=S.F32, device=dev))
        elif epi == "resid":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=S.F32, device=dev), resid=torch.randn(M, N, device=dev),
                      out2=torch.empty(M, N, dtype=S.BF16, device=dev))
        elif epi == "gelu":
            kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=S.BF16, device=dev), act=L.MRG_ACT_GELU_SAVE_GRAD,
                      aux_out=torch.empty(M, N, dtype=torch.float16, device=dev))
        elif epi == "mulaux":
            kw.update(out=torch.empty(M, N, dtype=S.BF16, device=dev), act=L.MRG_ACT_MUL_AUX_F16, aux_in=torch.rand(M, N, device=dev).half())
        sets.append((A, B, kw))

    def runner(variant):
        def f(i):
            A, B, kw = sets[i % nrot]
            if variant == "vendor":
                torch.mm(A, B.t() if mode == "fwd" else B)
                return
            K.TUNE["force"] = variant
            try:
                K.gemm(A, B, M, N, Kd, **kw)
            finally:
                K.TUNE["force"] = None
        return f

    fns = {"vendor": runner("vendor"), "tuned": runner(None)}
    for v in variants:
        fns[NAMES.get(v, str(v))] = runner(v)
    graphs = {}
    for k, f in fns.items():
        try:
            graphs[k] = S.graph_of(f, reps)
        except Exception as e:
            print(f"  {k}: {type(e).__name__}", file=sys.stderr)
    res = {k: [] for k in graphs}
    for _ in range(rounds):
        for k, g in graphs.items():
            res[k].append(S.time_graph(g, reps))
    fl = 2.0 * M * N * Kd
    line = f"{tag:14s} {mode:5s} {epi:8s} {M:6d}x{N:5d}x{Kd:5d} "
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"| {k} {m:5.1f} us {fl / m / 1e6:4.0f} TF "
    print(line, flush=True)


if __name__ == "__main__":
    Mq, Mc = 16384, 8256          # C2: 64 x 256 query tokens, 64 x 129 context tokens
    # an empty-ish launch for scale: a 128 x 128 x 64 product
    case("tiny", "fwd", "plain", 128, 128, 64)
    case("proj_context", "fwd", "f32", Mc, 256, 768)
    case("q", "fwd", "bias", Mq, 256, 256)
    case("kv", "fwd", "bias", Mc, 512, 256)
    case("attn proj", "fwd", "f32", Mq, 256, 256)
    case("fc1", "fwd", "gelu", Mq, 1024, 256)
    case("fc2", "fwd", "resid", Mq, 256, 1024)
    case("blk qkv", "fwd", "bias", Mq, 768, 256)
    case("out_proj", "fwd", "bias", Mq, 1024, 256)
    case("fc2 dgrad", "dgrad", "mulaux", Mq, 1024, 256)
    case("fc1 dgrad", "dgrad", "plain", Mq, 256, 1024)
    case("proj dgrad", "dgrad", "plain", Mq, 256, 256)
    case("qkv dgrad", "dgrad", "plainf32", Mq, 256, 768)
