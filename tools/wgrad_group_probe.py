#!/usr/bin/env python3
"""Where the time of a grouped weight-gradient launch goes: the stream-K kernel with grid = tiles (every workgroup owns one whole tile, as
in gemm_p8_group_kernel) and its per-workgroup trace: start, main loop (wall + shader clock), epilogue -- by round and by problem.
usage: wgrad_group_probe.py [c2|c4] [blocks] [randn|zeros]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mirage_amd import kernels as K, _lib as L

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fill = sys.argv[3] if len(sys.argv) > 3 else "randn"
D, T = (768, 8256) if which == "c2" else (1024, 6336)
torch.manual_seed(0)
ts, names = [], []
for b in range(nb):
    for nm, (M, N) in zip(("qkv", "proj", "fc1", "fc2"), ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D))):
        A = torch.randn(T, M, device=dev); B = torch.randn(T, N, device=dev)
        if fill == "zeros": A.zero_(); B.zero_()
        ts.append((A.bfloat16(), B.bfloat16(), M, N, T, M, N, torch.empty(M, N, device=dev), N, 1)); names.append(nm)
tile_prob = []
for i, t in enumerate(ts):
    tile_prob += [names[i]] * (((t[2] + 255) // 256) * ((t[3] + 255) // 256))
tiles = len(tile_prob)


def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    r = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        r.append(s.elapsed_time(e) * 1e3)
    return float(np.median(r))


fl = sum(2.0 * t[2] * t[3] * T for t in ts)
a = timed(lambda: K.wgrad_group(ts, T, streamk=False))
K.SK_GRID = tiles
b = timed(lambda: K.wgrad_group(ts, T, streamk=True))
print(f"{which} {nb} blocks, {fill}: {tiles} tiles, K = {T} ({T // 64} k-tiles); gemm_p8_group_kernel {a:.1f} us ({fl / a / 1e6:.0f} TF/s); "
      f"stream-K kernel with grid = tiles {b:.1f} us")
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
L.lib.mrg_debug_sk_trace(buf.data_ptr())
K.wgrad_group(ts, T, streamk=True); torch.cuda.synchronize()
L.lib.mrg_debug_sk_trace(None)
t = buf.cpu().numpy().reshape(-1, 32)
u = np.nonzero(t[:, 0])[0]
t0 = t[u, 0].min()
start = (t[u, 0] - t0) / 100.0
main = (t[u, 1] - t[u, 0]) / 100.0
mhz = (t[u, 31] - t[u, 30]) / (t[u, 1] - t[u, 0]).astype(np.float64) * 100.0
epi = (t[u, 4] - t[u, 2]) / 100.0
end = (t[u, 4] - t0) / 100.0
tile = (t[u, 3] & 0xffffffff).astype(int)
first = start < 5.0
print(f"workgroups traced {len(u)}; launch ends at {end.max():.1f} us")
for nm, sel in (("first round (started < 5 us)", first), ("later", ~first)):
    if sel.sum():
        print(f"  {nm}: {sel.sum()} workgroups, start {np.median(start[sel]):.1f} us, main loop {np.median(main[sel]):.1f} us "
              f"({np.median(main[sel]) / (T // 64):.3f} per k-tile) at {np.median(mhz[sel]):.0f} MHz, epilogue {np.median(epi[sel]):.1f} us, end {np.median(end[sel]):.1f} (max {end[sel].max():.1f})")
for nm in ("qkv", "proj", "fc1", "fc2"):
    sel = np.array([tile_prob[x] == nm for x in tile]) & first
    if sel.sum():
        print(f"  first round, {nm}: {sel.sum()} tiles, main loop {np.median(main[sel]):.1f} us (min {main[sel].min():.1f} max {main[sel].max():.1f}), epilogue {np.median(epi[sel]):.1f}")
order = np.argsort(start)
late = order[first.sum():][:8]
print("  first later starts:", ", ".join(f"{start[i]:.1f}" for i in late))
