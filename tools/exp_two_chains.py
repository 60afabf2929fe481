"""Feasibility experiment (timing only -- the two chains race on the gradient buffers): the C2 step as two half-batch
forward+backward chains on two streams inside one hipGraph, versus the single full-batch chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mirage_amd import builder, optim
from mirage_amd.trainer import TrainStep
dev = torch.device("cuda:0")
doms = ["bscan", "slo"]; B = 64; Kk = 128
args = builder.make_args(doms, doms)
model = builder.get_model(args, "miragepre_base").to(dev).train()
loss_fns = builder.get_losses(args)
sched = [1e-4] * 64
opt = optim.FusedAdamW(model, sched, [0.05] * 64, betas=(0.9, 0.95), eps=1e-8)
def synth(b):
    rng = np.random.default_rng(1)
    return {d: torch.from_numpy(rng.random((b, 1, 512, 512), dtype=np.float32)).to(dev) for d in doms}
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
# baseline: one chain
ts = TrainStep(model, loss_fns, opt, synth(B), Kk, use_graph=True)
ts.load(synth(B)); ts.warmup_and_capture(2)
print("one chain  B=64: %.3f ms/step" % timeit(lambda: ts.step()))
# two chains
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
halves = [TrainStep(model, loss_fns, opt, synth(B // n), Kk, use_graph=False) for _ in range(n)]
for h in halves: h.load(synth(B // n))
streams = [torch.cuda.Stream() for _ in range(n)]
def body():
    cur = torch.cuda.current_stream()
    for h, s in zip(halves, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            h._fwd_bwd()
    for s in streams: cur.wait_stream(s)
    halves[0]._opt()
s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s0):
    for _ in range(2): body()
torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
print("%d chains B=%d each: %.3f ms/step" % (n, B // n, timeit(lambda: g.replay())))
