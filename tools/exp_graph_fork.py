#!/usr/bin/env python3
"""When does the second branch of a forked hipGraph start?  PRE kernels on the capture stream, then two independent chains of N
kernels (one on a lane stream, one on the capture stream), join, POST kernels.  Device timestamps (mrg_debug_timestamp) at the start
of each chain tell the start lag of the lane chain relative to the main chain inside a replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirage_amd import kernels as K

dev = torch.device("cuda", 0)
M, D = 16384, 256
x = [torch.randn(M, D, device=dev).bfloat16() for _ in range(3)]
w = [(torch.randn(D, D, device=dev) * 0.05).bfloat16() for _ in range(3)]
b = torch.zeros(D, device=dev)
buf = [[torch.empty(M, D, dtype=torch.bfloat16, device=dev) for _ in range(2)] for _ in range(3)]
big_a = torch.randn(8256, 768, device=dev).bfloat16(); big_w = (torch.randn(3072, 768, device=dev) * 0.02).bfloat16()
big_o = torch.empty(8256, 3072, dtype=torch.bfloat16, device=dev)


def chain(i, n):
    a = x[i]
    for k in range(n):
        K.gemm(a, w[i], M, D, D, bias=b, out=buf[i][k & 1])
        a = buf[i][k & 1]


def pre(n):
    for _ in range(n):
        K.gemm(big_a, big_w, 8256, 3072, 768, bias=torch_bias, out=big_o)


torch_bias = torch.zeros(3072, device=dev)
lane = torch.cuda.Stream()
stamps = torch.zeros(8, dtype=torch.int64, device=dev)


def stamp(i):
    K.check(K.lib.mrg_debug_timestamp(stamps.data_ptr() + 8 * i, K._s()), "mrg_debug_timestamp")     # writes the device wall clock


def build(npre, n, order):
    def fn():
        pre(npre)
        main = torch.cuda.current_stream(); ev = torch.cuda.Event(); ev.record(main)
        def lane_part():
            lane.wait_event(ev)
            with torch.cuda.stream(lane):
                stamp(1); chain(1, n); stamp(3)
        def main_part():
            stamp(0); chain(0, n); stamp(2)
        if order == "lane_first":
            lane_part(); main_part()
        else:
            main_part(); lane_part()
        main.wait_stream(lane)
        chain(2, 2)
    return fn


for npre in (0, 40, 120):
    for n in (12, 36):
        for order in ("lane_first", "main_first"):
            fn = build(npre, n, order)
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            lags, spans = [], []
            for _ in range(5):
                g.replay(); torch.cuda.synchronize()
                s = stamps.tolist()
                lags.append((s[1] - s[0]) / 100.0); spans.append((max(s[2], s[3]) - min(s[0], s[1])) / 100.0)
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) / 20 * 1e6
            print(f"pre {npre:3d} n {n:2d} {order:10s}: lane start lag {sorted(lags)[2]:8.1f} us, fork->join span {sorted(spans)[2]:8.1f} us, replay {tot:8.1f} us", flush=True)
