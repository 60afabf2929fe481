#!/usr/bin/env python3
"""Offer the 64-row GEMM variant (MRG_GEMM_GLDS64) to the table entries a configuration uses: re-run the isolated autotuner for the
eligible keys (A k-contiguous, no split-K), accept only switches TO the new variant, then A/B the whole step (hipGraph replay) old
table vs new table and keep the new one if it is not slower.  Usage: retune_v8.py c2|c3|c4|c5 [--save]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from mirage_amd import builder, kernels as K, optim, _lib as L
from mirage_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, doms)
torch.manual_seed(0)
model = builder.get_model(args, model_name).to(dev).train()
pool = Bn.synth_batches(doms, B, 2, 1234, dev)
lrs = optim.cosine_scheduler(1e-4, 0.0, 1, 4096, warmup_epochs=1, warmup_steps=100)
wds = optim.cosine_scheduler(0.05, 0.05, 1, 4096)
opt = optim.FusedAdamW(model, lrs, wds, betas=(0.9, 0.95), eps=1e-8)
opt.arena.refresh_shadow(force=True)
loss_fns = builder.get_losses(args)


def measure(reps=4, steps=15):
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


K.TUNE["hits"] = set()
base = measure()
hits = set(K.TUNE["hits"]); K.TUNE["hits"] = None
old = dict(K._tune_cache)
elig = [k for k in hits if k[0] == L.MRG_OP_KCONTIG and not k[8] and old.get(k, (0, 1))[1] == 1]
for k in elig:
    K._tune_cache.pop(k, None)
K.TUNE["log"].clear()
measure(reps=1, steps=2)                         # the eager warm-up inside re-times the popped keys (all candidates)
switched = []
for k in elig:
    new = K._tune_cache.get(k)
    if new is None or new[0] != L.MRG_GEMM_GLDS64:
        K._tune_cache[k] = old[k]
    else:
        switched.append(k)
for k in sorted(switched, key=lambda k: -(k[2] * k[3] * k[4])):
    print(f"  {k[:6]} {old[k]} -> {K._tune_cache[k]}")
new_t = measure()
base2 = None
if switched:
    saved = dict(K._tune_cache)
    K._tune_cache.clear(); K._tune_cache.update(old)
    base2 = measure()
    K._tune_cache.clear(); K._tune_cache.update(saved)
    new_t = min(new_t, measure())
print(f"{cfg}: old table {base:.3f} / {base2} ms, with {len(switched)} entries on the 64-row kernel {new_t:.3f} ms")
ref = min(base, base2) if base2 else base
if "--save" in sys.argv and switched and new_t <= ref:
    K.save_tune_cache(); print("saved")
elif switched and new_t > ref:
    print("not saved (no gain in step)")
