#!/usr/bin/env python3
"""Phase timestamps of the fused post-attention chain (debug build of csrc/rowfuse.hip with -DMRG_RF_TS, loaded through MRG_LIB_PATH):
cycles since the workgroup's start at every phase boundary, for a few workgroups."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K

dev = "cuda:0"
names = {0: "start", 1: "prologue landed", 2: "stage-0 tiles", 3: "x = .. + bias (+ stage)", 4: "stats 1 + x copy", 5: "LN written", 22: "last tile",
         23: "y staged", 24: "y copy issued", 25: "stores drained"}
for c in range(4):
    names[6 + 4 * c] = f"chunk {c}: fc1 tiles"; names[7 + 4 * c] = f"chunk {c}: gelu epilogue"
    names[8 + 4 * c] = f"chunk {c}: fc2 k<=3"; names[9 + 4 * c] = f"chunk {c}: fc2 tiles"


def lin(nout, nin):
    l = types.SimpleNamespace(); l.w16 = (torch.randn(nout, nin, device=dev) * nin ** -0.5).bfloat16(); l.bias = torch.randn(nout, device=dev) * 0.1
    return l


for M in (16384, 6336):
    n = types.SimpleNamespace(weight=torch.ones(256, device=dev), bias=torch.zeros(256, device=dev), eps=1e-6)
    proj, fc1, fc2 = lin(256, 256), lin(1024, 256), lin(256, 1024)
    o = torch.randn(M, 256, device=dev).bfloat16(); r = torch.randn(M, 256, device=dev)
    for _ in range(3):
        y32, y16, saved = K.rowfuse_mlp(o, proj, proj.w16, r, n, fc1, fc1.w16, fc2, fc2.w16, True, True)
    torch.cuda.synchronize()
    ts = saved[3].view(-1, 64).cpu()
    print(f"M = {M}: {M // 64} workgroups; cycles since start (workgroup 0 | median over workgroups | delta of the median)")
    med = ts.median(0).values
    prev = 0.0
    for i in sorted(names):
        print(f"  {i:2d} {names[i]:28s} {ts[0, i]:9.0f} {med[i]:9.0f} {med[i] - prev:8.0f}")
        prev = float(med[i])
