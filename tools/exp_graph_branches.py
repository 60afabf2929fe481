#!/usr/bin/env python3
"""How does hipGraph (torch.cuda.CUDAGraph on ROCm 7) execute two independent captured branches?  Two chains of n small
kernels (each ~15-25 us, under-filling the GPU like the output-adapter launches) forked from / joined into the capture stream;
variants: which branch is captured first, one or two side streams, chain length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K

dev = torch.device("cuda", 0)
M, D = 16384, 256
x = [torch.randn(M, D, device=dev).bfloat16() for _ in range(2)]
w = [(torch.randn(D, D, device=dev) * 0.05).bfloat16() for _ in range(2)]
b = torch.zeros(D, device=dev)
buf = [[torch.empty(M, D, dtype=torch.bfloat16, device=dev) for _ in range(2)] for _ in range(2)]


def chain(i, n):
    a = x[i]
    for k in range(n):
        K.gemm(a, w[i], M, D, D, bias=b, out=buf[i][k & 1])
        a = buf[i][k & 1]


def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for n in (10, 30, 60):
    def serial():
        chain(0, n); chain(1, n)
    def side_first():          # branch on the side stream captured first, then the main-stream branch
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        s1.wait_event(ev)
        with torch.cuda.stream(s1): chain(1, n)
        chain(0, n)
        main.wait_stream(s1)
    def main_first():
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        chain(0, n)
        s1.wait_event(ev)
        with torch.cuda.stream(s1): chain(1, n)
        main.wait_stream(s1)
    def both_side():
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        for i, st in enumerate((s1, s2)):
            st.wait_event(ev)
            with torch.cuda.stream(st): chain(i, n)
        main.wait_stream(s1); main.wait_stream(s2)
    def interleaved():          # launches alternate between the two streams (creation order a0 b0 a1 b1 ...)
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        s1.wait_event(ev)
        a = [x[0], x[1]]
        for k in range(n):
            K.gemm(a[0], w[0], M, D, D, bias=b, out=buf[0][k & 1]); a[0] = buf[0][k & 1]
            with torch.cuda.stream(s1):
                K.gemm(a[1], w[1], M, D, D, bias=b, out=buf[1][k & 1]); a[1] = buf[1][k & 1]
        main.wait_stream(s1)
    res = {f.__name__: timeit(f) for f in (serial, side_first, main_first, both_side, interleaved)}
    one = timeit(lambda: chain(0, n))
    print(f"n={n:3d}: one chain {one:7.1f} us | " + " | ".join(f"{k} {v:7.1f}" for k, v in res.items()), flush=True)
