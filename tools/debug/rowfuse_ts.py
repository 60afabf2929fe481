#!/usr/bin/env python3
"""Phase timestamps of the fused post-attention chain (debug build of csrc/rowfuse.hip with -DMRG_RF_TS, loaded through MRG_LIB_PATH):
cycles since the workgroup's start at every phase boundary, median over the workgroups.  M % 128 == 0: the row-owner-wave kernel
(rowchain_fwd_kernel, 128 rows per workgroup); MRG_ROWCHAIN=0 or other M: the first design (64 rows per workgroup)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K

dev = "cuda:0"
chain = os.environ.get("MRG_ROWCHAIN", "1") != "0"


def names_for(M):
    if chain and M % 128 == 0:
        n = {0: "start", 1: "prologue landed", 2: "stage-0 tiles", 3: "x = .. + bias, x out", 5: "LayerNorm, ln out", 22: "last tile", 31: "y copies issued", 32: "stores drained"}
        for c in range(8):
            n[6 + 3 * c] = f"chunk {c}: fc1 tiles"; n[7 + 3 * c] = f"chunk {c}: gelu + fc2 tiles"; n[8 + 3 * c] = f"chunk {c}: h / h' copies"
        return n, 128
    n = {0: "start", 1: "prologue landed", 2: "stage-0 tiles", 3: "x = .. + bias (+ stage)", 4: "stats 1 + x copy", 5: "LN written", 22: "last tile",
         23: "y staged", 24: "y copy issued", 25: "stores drained"}
    for c in range(4):
        n[6 + 4 * c] = f"chunk {c}: fc1 tiles"; n[7 + 4 * c] = f"chunk {c}: gelu epilogue"
        n[8 + 4 * c] = f"chunk {c}: fc2 k<=3"; n[9 + 4 * c] = f"chunk {c}: fc2 tiles"
    return n, 64


def lin(nout, nin):
    l = types.SimpleNamespace(); l.w16 = (torch.randn(nout, nin, device=dev) * nin ** -0.5).bfloat16(); l.bias = torch.randn(nout, device=dev) * 0.1
    return l


for M in (16384, 6336):
    names, rows = names_for(M)
    n = types.SimpleNamespace(weight=torch.ones(256, device=dev), bias=torch.zeros(256, device=dev), eps=1e-6)
    proj, fc1, fc2 = lin(256, 256), lin(1024, 256), lin(256, 1024)
    o = torch.randn(M, 256, device=dev).bfloat16(); r = torch.randn(M, 256, device=dev)
    for _ in range(3):
        y32, y16, saved = K.rowfuse_mlp(o, proj, proj.w16, r, n, fc1, fc1.w16, fc2, fc2.w16, True, True)
    torch.cuda.synchronize()
    ts = saved[3].view(-1, rows)[:, :64].cpu()
    print(f"M = {M}: {M // rows} workgroups of {rows} rows; cycles since start (workgroup 0 | median over workgroups | delta of the median)")
    med = ts.median(0).values
    prev = 0.0
    for i in sorted(names):
        print(f"  {i:2d} {names[i]:28s} {ts[0, i]:9.0f} {med[i]:9.0f} {med[i] - prev:8.0f}")
        prev = float(med[i])
