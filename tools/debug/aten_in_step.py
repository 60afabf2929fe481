#!/usr/bin/env python3
"""Which torch (aten) operators still launch device work inside one training step, and from where (the ~27
copyBuffer and ~5 aten fill / add launches per replayed C2 step).  Eager step under torch.profiler with Python stacks."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench as Bn
from mirage_amd import builder, optim
from mirage_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
model_name, doms, B, Kk, train = Bn.CONFIGS[cfg]
dev = torch.device("cuda", 0)
args = builder.make_args(doms, doms)
torch.manual_seed(0)
model = builder.get_model(args, model_name).to(dev).train()
pool = Bn.synth_batches(doms, B, 2, 1234, dev)
opt = optim.FusedAdamW(model, [1e-4] * 64, [0.05] * 64, betas=(0.9, 0.95), eps=1e-8)
ts = TrainStep(model, builder.get_losses(args), opt, pool[0], Kk, use_graph=False)
ts.load(pool[0])
for _ in range(2):
    ts.step(pool[1])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step(pool[0])
    torch.cuda.synchronize()
rows = collections.Counter()
want = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::ones_like", "aten::zeros")
for ev in prof.events():
    if ev.name not in want:
        continue
    site = next((s for s in (ev.stack or []) if "mirage_amd" in s or "bench.py" in s), "(autograd engine / torch internals)")
    rows[(ev.name, site.split("/root/repo/")[-1][:120], str(ev.input_shapes[:1])[:40])] += 1
for (name, site, shp), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:18s} {shp:40s} {site}")
print("total:", sum(rows.values()))
