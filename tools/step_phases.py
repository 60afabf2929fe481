#!/usr/bin/env python3
"""Phase structure of a graph-replayed training step from timestamps recorded INSIDE the graph (mrg_debug_timestamp: the device's
100 MHz wall clock when a stream reaches a point), i.e. without a profiler on the host side.  Usage: MRG_TS=1 python tools/step_phases.py [c2|c3|c4]"""
import os, sys
os.environ["MRG_TS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
from mirage_amd import builder, kernels as K, optim
from mirage_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, doms)
torch.manual_seed(0)
model = builder.get_model(args, model_name).to(dev).train()
pool = Bn.synth_batches(doms, B, 2, 1234, dev)
opt = optim.FusedAdamW(model, [1e-4] * 4096, [0.05] * 4096, betas=(0.9, 0.95), eps=1e-8)
ts = TrainStep(model, builder.get_losses(args), opt, pool[0], Kk, use_graph=True)
ts.load(pool[0]); ts.warmup_and_capture(1)
acc = {}
# "pipelined" (default): several replays are queued before each read-out, as in bench.py's timed loop, so the host's graph submission
# runs ahead of the GPU; "sync": one replay at a time (the GPU starts beside the host's submission: parallel branches then start
# in submission order, ~3 us per node apart -- tools/exp_graph_fork.py)
mode = sys.argv[2] if len(sys.argv) > 2 else "pipelined"
for it in range(12):
    if mode == "sync":
        ts.step(pool[it & 1]); torch.cuda.synchronize()
    else:
        for _ in range(4):
            ts.step(pool[it & 1])
        torch.cuda.synchronize()
    if it < 2:
        continue
    t = K.TS["buf"].cpu().numpy()[:len(K.TS["names"])].astype(np.int64)
    t0 = t[K.TS["names"].index("step_start")]
    for n, v in zip(K.TS["names"], t):
        acc.setdefault(n, []).append((v - t0) * 0.01)        # 100 MHz -> us
print(f"{cfg} [{mode}]: phase timestamps inside one graph replay (us from step start, median of 10 replays)")
for n, v in sorted(acc.items(), key=lambda kv: np.median(kv[1])):
    print(f"  {np.median(v):9.1f}  {n}")
