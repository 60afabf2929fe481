#!/usr/bin/env python3
"""C5 (Large encoder-only forward, B = 256) with the encoder GEMM table entries overridden: one chain vs two chains, 8-phase one tile per
workgroup vs stream-K.  usage: c5_variants.py"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from mirage_amd import builder, kernels as K, engine as E, _lib as L

model_name, doms, B, Kk, train = Bn.CONFIGS["c5"]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, ())
torch.manual_seed(0)
model = builder.get_model(args, model_name, with_decoders=False).to(dev).eval()
pool = Bn.synth_batches(doms, B, 2, 1234, dev)


def measure(chains, variant):
    E.TWO_CHAINS = chains
    saved = dict(K._tune_cache)
    for key in list(K._tune_cache):
        if key[2] in (65792, 32896) and key[4] % 64 == 0 and key[3] >= 1024:
            K._tune_cache[key] = (variant, 1)
    for M in (65792, 32896):
        for (N, Kd, act, res, f32) in ((3072, 1024, 0, False, False), (1024, 1024, 0, True, True), (4096, 1024, 1, False, False), (1024, 4096, 0, True, True)):
            K._tune_cache[(0, 0, M, N, Kd, act, res, f32, False)] = (variant, 1)
    static = {d: pool[0][d].clone() for d in doms}
    fwd = lambda: model(static, mask_inputs=False)[0]
    with torch.no_grad():
        fwd(); fwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
    best = 1e9
    for _ in range(3):
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
    del g; gc.collect(); torch.cuda.empty_cache()
    K._tune_cache.clear(); K._tune_cache.update(saved)
    return best


for chains in (1, 0):
    for v, name in ((L.MRG_GEMM_P8, "8-phase"), (L.MRG_GEMM_P8SK, "stream-K")):
        print(f"C5 forward, {'two chains' if chains else 'one chain'}, encoder GEMMs on {name}: {measure(chains, v):.2f} ms", flush=True)
