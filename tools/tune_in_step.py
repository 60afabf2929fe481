#!/usr/bin/env python3
"""Coordinate-descent refinement of the GEMM tuning table against the WHOLE training step (hipGraph replay) instead of isolated
launches: inside the step the dgrad and wgrad GEMMs share the GPU, so the isolated optimum is not always the in-step optimum.
Usage: tune_in_step.py c2|c3|c4 [--save]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
from mirage_amd import builder, kernels as K, optim, _lib as L
from mirage_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, doms)
torch.manual_seed(0)
if not train:
    args = builder.make_args(doms, ())
model = builder.get_model(args, model_name, with_decoders=train).to(dev).train(train)
pool = Bn.synth_batches(doms, B, 2, 1234, dev)
if train:
    lrs = optim.cosine_scheduler(1e-4, 0.0, 1, 4096, warmup_epochs=1, warmup_steps=100)
    wds = optim.cosine_scheduler(0.05, 0.05, 1, 4096)
    opt = optim.FusedAdamW(model, lrs, wds, betas=(0.9, 0.95), eps=1e-8)
    opt.arena.refresh_shadow(force=True)
    loss_fns = builder.get_losses(args)


def measure_forward(reps=3, steps=6):
    """encoder-only forward (C5) replayed from a hipGraph, as bench.py runs it"""
    static = {d: pool[0][d].clone() for d in doms}
    fwd = lambda: model(static, mask_inputs=False)[0]
    with torch.no_grad():
        fwd(); fwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
    best = 1e9
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): g.replay()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    del g; gc.collect(); torch.cuda.empty_cache()
    return best


def measure(reps=3, steps=15):
    if not train:
        return measure_forward(reps, max(4, steps // 3))
    ts = TrainStep(model, loss_fns, opt, pool[0], Kk, use_graph=True)
    ts.load(pool[0]); ts.warmup_and_capture(1)
    best = 1e9
    for _ in range(reps):
        for _ in range(3): ts.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): ts.step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    del ts; gc.collect(); torch.cuda.empty_cache()
    return best


# the first measurement also records which table entries this configuration uses (and tunes missing ones in isolation)
K.TUNE["hits"] = set()
base = measure()
hits = set(K.TUNE["hits"]); K.TUNE["hits"] = None
print(f"baseline {base:.3f} ms/step")
keys = sorted(hits, key=lambda k: -(2.0 * k[2] * k[3] * k[4]))
used = [k for k in keys if 2.0 * k[2] * k[3] * k[4] > float(os.environ.get('MRG_TUNE_MIN_FLOP', '5e9'))][:int(os.environ.get("MRG_TUNE_TOP", "16"))]
cands = (L.MRG_GEMM_GLDS3, L.MRG_GEMM_GLDS4, L.MRG_GEMM_PIPE128, L.MRG_GEMM_WS256, L.MRG_GEMM_P8)
if os.environ.get("MRG_TUNE_ONLY"):
    cands = tuple(int(x) for x in os.environ["MRG_TUNE_ONLY"].split(","))
cur = base
for key in used:
    v0, s0 = K._tune_cache[key]
    trials = [(v, s0) for v in cands if v != v0 and not (v in (L.MRG_GEMM_WS256, L.MRG_GEMM_P8) and key[2] <= 128) and not (v == L.MRG_GEMM_P8 and key[4] % 64)
              and not (v == L.MRG_GEMM_P8 and (key[3] <= 128 or ((key[2] + 255) // 256) * ((key[3] + 255) // 256) * s0 < 192))]
    if key[8]:                                   # split-K tunable (wgrad): also neighbours of the split factor
        trials += [(v0, s) for s in (max(1, s0 // 2), s0 * 2) if s != s0 and (key[4] // 64) // s >= 4]
    best_choice, best_t = (v0, s0), cur
    for ch in trials:
        K._tune_cache[key] = ch
        try:
            t = measure(reps=2, steps=12)
        except Exception as e:
            print("  failed", key, ch, e); continue
        if os.environ.get("MRG_TUNE_VERBOSE"):
            print(f"    trial {key[:5]} act={key[5]} {ch}: {t:.3f} ms (current {cur:.3f})", flush=True)
        if t < best_t * 0.997:
            best_choice, best_t = ch, t
    K._tune_cache[key] = best_choice
    if best_choice != (v0, s0):
        cur = measure()                          # re-measure with the full protocol before accepting
        if cur > best_t * 1.01 and cur > base:
            K._tune_cache[key] = (v0, s0); cur = measure()
    print(f"{key[:5]} act={key[5]} split={key[8]}: {(v0, s0)} -> {K._tune_cache[key]}  step {cur:.3f} ms", flush=True)
print(f"final {cur:.3f} ms/step (baseline {base:.3f})")
if "--save" in sys.argv and cur < base:
    K.save_tune_cache()
    print("saved")
