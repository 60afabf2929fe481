#!/usr/bin/env python3
"""Localises the NaN of the reverted two-chain backward (tools/debug/patches/two_chain_backward.patch).  The patch applies at
commit 4fbbead:  git worktree add .ab_nan 4fbbead && cd .ab_nan && git apply ../tools/debug/patches/two_chain_backward.patch &&
make -C mirage_amd/csrc && cp ../tools/debug/two_chain_bwd_nan_hunt.py . && python two_chain_bwd_nan_hunt.py
Result (MI355X, C4 shapes, B = 16, depth 24; profiles/r04_experiments.md): eager two-chain backward == single chain bit for bit, also with
every allocation poisoned with NaN; only the hipGraph-captured step is wrong, and only in what chain B's norm2 LayerNorm-backward launch
hands to the side stream (bf16 twin of dx_mid -> attn.proj weight gradient; its partial rows -> norm2 weight / bias, fc2 bias): a
capture-time ordering / lifetime defect of the patch's half-tensor hand-off, not of kernels.Deferred / mrg_reduce_batch / the partial
column sums the product shares with it (tests/test_fullsize_gpu.py::test_poisoned_workspaces_leave_every_gradient_unchanged)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
import torch
from mirage_amd import builder, optim, engine as E, kernels as K
from mirage_amd.trainer import TrainStep
from oracle import mirage_oracle as O
import test_model_gpu as T
DEV = "cuda:0"
depth = int(os.environ.get("DEPTH", "24")); B = int(os.environ.get("B", "16"))
cfg = O.make_config("large", ("bscan", "slo", "bscanlayermap"), depth=depth)
Kk = 98
x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=1234).items()}
(p, u, g), _, _, _ = O.synthetic_masks(cfg, B, Kk, seed=99)
noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
def run(bwd2, graph):
    E.TWO_CHAINS_BWD = bwd2
    model, args, _ = T._build(cfg)
    model.train()
    opt = optim.FusedAdamW(model, [1e-4] * 8, [0.05] * 8, betas=(0.9, 0.95), eps=1e-8)
    ts = TrainStep(model, builder.get_losses(args), opt, x, Kk, use_graph=graph, feed_mask_noise=True)
    ts.load(x)
    for dst, src in zip(ts.static_noise, noise):
        dst.copy_(src)
    if graph:
        ts.warmup_and_capture(warm_iters=1)
    ts.step(x, noise=noise)
    torch.cuda.synchronize()
    return {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
ref = run(0, True)
for bwd2, graph in ((1, False), (1, True)):
    gr = run(bwd2, graph)
    bad = {n: int((~torch.isfinite(v)).sum()) for n, v in gr.items() if not torch.isfinite(v).all()}
    off = sorted(((float((gr[n].float() - ref[n].float()).norm() / (ref[n].float().norm() + 1e-20)), n) for n in ref if n not in bad), reverse=True)[:6]
    print(f"two-chain backward {bwd2} graph {graph}: non-finite {bad if bad else 'none'}; largest relative differences vs single chain: {[(round(a, 4), n) for a, n in off]}", flush=True)
