import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from mirage_amd import builder, optim, engine as E
from mirage_amd.trainer import TrainStep
from oracle import mirage_oracle as O
import test_model_gpu as T
DEV = "cuda:0"
cfg = O.make_config("base", ("bscan", "slo"))
B, Kk = 16, 128
x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=5).items()}
(p, u, g), tm, ik, ir = O.synthetic_masks(cfg, B, Kk, seed=6)
noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
def run(side, graph, steps=3):
    E._Side.enabled = side
    model, args, _ = T._build(cfg); model.train()
    opt = optim.FusedAdamW(model, [1e-4] * 8, [0.05] * 8)
    ts = TrainStep(model, builder.get_losses(args), opt, x, Kk, use_graph=graph, feed_mask_noise=True)
    ts.load(x)
    for dst, src in zip(ts.static_noise, noise): dst.copy_(src)
    ts.warmup_and_capture(1)
    losses = []
    for _ in range(steps):
        ts.step(x, noise=noise); losses.append(float(ts.loss))
    torch.cuda.synchronize()
    return losses, opt.arena.flat_g.clone(), opt.arena.flat_p.clone(), float(opt.grad_norm)
import itertools
res = {}
for name, side, graph in (("side graph", True, True), ("side eager", True, False), ("noside graph", False, True), ("noside eager", False, False), ("noside eager 2", False, False)):
    res[name] = run(side, graph, steps=int(os.environ.get("STEPS", "1")))
    print(name, res[name][0], res[name][3])
for a, b in itertools.combinations(res, 2):
    ga, gb = res[a][1], res[b][1]
    print(f"{a:15s} vs {b:15s}: g rel {float((ga - gb).norm() / ga.norm()):.3e}  p max abs {float((res[a][2] - res[b][2]).abs().max()):.3e}")
