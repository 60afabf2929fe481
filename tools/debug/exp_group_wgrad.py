"""Emulation of a grouped weight-gradient launch: one 8-phase 256x256 launch WITHOUT split-K over as many output tiles as several
encoder blocks' weight gradients have together (contraction over all 8256 / 6336 tokens per tile), against the per-GEMM tuned split-K path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = "cuda:0"
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
def run(T, D, nblk, tag):
    # tiles per block (256 x 256): qkv 3D x D, proj D x D, fc1 4D x D, fc2 D x 4D  -> emulate as ONE tall problem [rows x D] with K = T
    rows = nblk * (3 * D + D + 4 * D + 4 * D)          # fc2 (D x 4D) has the same tile count as fc1
    dy = torch.randn(T, rows, device=dev).bfloat16(); x = torch.randn(T, D, device=dev).bfloat16()
    dW = torch.empty(rows, D, device=dev)
    K.TUNE["enabled"] = False
    K.TUNE["force"] = L.MRG_GEMM_P8
    t = bench(lambda: K.gemm(dy, x, rows, D, T, a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED, out=dW, splitk=1))
    fl = 2.0 * rows * D * T
    tiles = ((rows + 255) // 256) * ((D + 255) // 256)
    print(f"{tag}: grouped-equivalent of {nblk} blocks: {tiles} tiles ({tiles / 256:.2f} rounds), {t * 1e6:8.1f} us = {t * 1e6 / nblk:6.1f} us/block, {fl / t / 1e12:6.0f} TF/s")
    K.TUNE["force"] = None
    K.TUNE["enabled"] = True
def run_old(T, D, tag):
    tot = 0.0
    for name, N, Kd in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        dy = torch.randn(T, N, device=dev).bfloat16(); x = torch.randn(T, Kd, device=dev).bfloat16(); dW = torch.empty(N, Kd, device=dev)
        sk = K.pick_splitk(N, Kd, T)
        t = bench(lambda: K.gemm(dy, x, N, Kd, T, a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED, out=dW, splitk=sk, tune_split=True))
        tot += t
    print(f"{tag}: per-GEMM tuned split-K path (incl. its reductions): {tot * 1e6:6.1f} us/block")
for T, D, tag in ((8256, 768, "C2"), (6336, 1024, "C4")):
    run_old(T, D, tag)
    for nblk in (1, 2, 3, 4, 6, 7, 12):
        run(T, D, nblk, tag)
