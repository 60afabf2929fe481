#!/usr/bin/env python3
"""Per-workgroup phase timeline of one stream-K launch (mrg_debug_sk_trace): where a range's time goes -- main loops, partial-image
exchange (store / finisher read), epilogues.  usage: sk_trace.py M N K [grid] [fwd|dgrad] [epi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mirage_amd import kernels as K, _lib as L

M, N, Kd = (int(x) for x in sys.argv[1:4])
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"
epi = sys.argv[6] if len(sys.argv) > 6 else "bias"
dev = "cuda:0"
torch.manual_seed(0)
A = torch.randn(M, Kd, device=dev).bfloat16()
B = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16() if mode == "fwd" else (torch.randn(Kd, N, device=dev) * 0.05).bfloat16()
kw = dict(b_mode=L.MRG_OP_KSTRIDED) if mode != "fwd" else {}
if epi == "bias":
    kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
elif epi == "gelu":
    kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, dtype=torch.bfloat16, device=dev), act=L.MRG_ACT_GELU_SAVE_GRAD,
              aux_out=torch.empty(M, N, dtype=torch.float16, device=dev))
elif epi == "resid":
    kw.update(bias=torch.randn(N, device=dev), out=torch.empty(M, N, device=dev), resid=torch.randn(M, N, device=dev))
K.TUNE["force"] = L.MRG_GEMM_P8SK
K.SK_GRID = grid
P = grid or 256
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
for _ in range(3):
    K.gemm(A, B, M, N, Kd, **kw)
torch.cuda.synchronize()
L.lib.mrg_debug_sk_trace(buf.data_ptr())
K.gemm(A, B, M, N, Kd, **kw)
torch.cuda.synchronize()
L.lib.mrg_debug_sk_trace(None)
t = buf.cpu().numpy().reshape(-1, 32)
used = np.nonzero(t[:, 0])[0]
t0 = t[used, 0].min()
print(f"{M}x{N}x{Kd} grid {len(used)}: start spread {(t[used, 0].max() - t0) / 100:.2f} us")
ends = []
tot = {"main": [], "exch": [], "epi": []}
rows = []
for w in used:
    r = t[w]
    prev = r[0]
    line = f"wg {w:3d} start {(r[0] - t0) / 100:6.2f}"
    m = x = e = 0.0
    for si in range(7):
        if r[1 + 4 * si] == 0:
            break
        ml, ex, meta, ep = r[1 + 4 * si], r[2 + 4 * si], r[3 + 4 * si], r[4 + 4 * si]
        role, tile, nkk = (meta >> 32) & 0xffff, meta & 0xffffffff, (meta >> 48) & 0xffff
        line += f" | t{tile:3d} k{nkk:2d} r{role} main {(ml - prev) / 100:5.2f} exch {(ex - ml) / 100:5.2f} epi {(ep - ex) / 100:5.2f}"
        m += (ml - prev) / 100; x += (ex - ml) / 100; e += (ep - ex) / 100
        prev = ep
    ends.append((prev - t0) / 100)
    tot["main"].append(m); tot["exch"].append(x); tot["epi"].append(e)
    rows.append(line + f" | end {(prev - t0) / 100:6.2f}")
for ln in rows[:12] + rows[len(rows) // 2: len(rows) // 2 + 6] + rows[-6:]:
    print(ln)
for w in used[:3]:
    e = t[w, 20:26]
    if e[0]:
        print(f"wg {w} epilogue stamps of its last segment (us, pass barrier / staged / own drain done): " + " ".join(f"{(x - e[0]) / 100:.2f}" for x in e))
print(f"end: min {min(ends):.2f} median {np.median(ends):.2f} max {max(ends):.2f} us;  per-WG mean main {np.mean(tot['main']):.2f} exch {np.mean(tot['exch']):.2f} "
      f"epi {np.mean(tot['epi']):.2f} us")
