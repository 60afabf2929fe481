#!/usr/bin/env python3
"""Grouped weight gradients (mrg_wgrad_group): one tile per workgroup against the stream-K schedule, isolated, C2 / C4 block shapes.
usage: bench_sk_group.py [c2|c4] [blocks ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
blocks_list = [int(x) for x in sys.argv[2:]] or [2, 4, 6]
D, T = (768, 8256) if which == "c2" else (1024, 6336)
torch.manual_seed(0)


def make(nblocks):
    ts = []
    for b in range(nblocks):
        for (M, N) in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
            A = torch.randn(T, M, device=dev); B = torch.randn(T, N, device=dev)
            fill = os.environ.get("MRG_BENCH_DATA", "randn")      # randn | small (x 0.02) | zeros | sign (|randn|): what bounds the launch? (DVFS)
            if fill == "zeros": A.zero_(); B.zero_()
            elif fill == "small": A *= 0.02; B *= 0.02
            elif fill == "sign": A.abs_(); B.abs_()
            A = A.bfloat16(); B = B.bfloat16()
            ts.append((A, B, M, N, T, M, N, torch.empty(M, N, device=dev), N, 1))
    return ts


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    r = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        r.append(s.elapsed_time(e) * 1e3)
    return float(np.median(r))


for nb in blocks_list:
    ts = make(nb)
    tiles = sum(((t[2] + 255) // 256) * ((t[3] + 255) // 256) for t in ts)
    fl = sum(2.0 * t[2] * t[3] * T for t in ts)
    a = timed(lambda: K.wgrad_group(ts, T, streamk=False))
    ref = [t[7].clone() for t in ts]
    b = timed(lambda: K.wgrad_group(ts, T, streamk=True))
    err = max(float((t[7] - r).abs().max() / r.abs().max()) for t, r in zip(ts, ref))
    print(f"{which} {nb} blocks: {tiles} tiles ({tiles / 256:.2f} rounds)  one-tile {a:7.1f} us {fl / a / 1e6:5.0f} TF | stream-K {b:7.1f} us {fl / b / 1e6:5.0f} TF | max rel diff {err:.1e}", flush=True)
    buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
    L.lib.mrg_debug_sk_trace(buf.data_ptr())
    K.wgrad_group(ts, T, streamk=True); torch.cuda.synchronize()
    L.lib.mrg_debug_sk_trace(None)
    t = buf.cpu().numpy().reshape(-1, 32)
    used = np.nonzero(t[:, 0])[0]
    t0 = t[used, 0].min()
    ends, mains, exch, epi = [], [], [], []
    for w in used:
        r = t[w]; prev = r[0]; m = x = e = 0
        for si in range(7):
            if r[1 + 4 * si] == 0: break
            m += r[1 + 4 * si] - prev; x += r[2 + 4 * si] - r[1 + 4 * si]; e += r[4 + 4 * si] - r[2 + 4 * si]; prev = r[4 + 4 * si]
        ends.append((prev - t0) / 100); mains.append(m / 100); exch.append(x / 100); epi.append(e / 100)
    print(f"    stream-K ranges end at min {min(ends):.1f} median {np.median(ends):.1f} max {max(ends):.1f} us; per range main {np.mean(mains):.1f} exchange {np.mean(exch):.1f} epilogue {np.mean(epi):.1f}")
