#!/usr/bin/env python3
"""Where does the SYSTEMATIC part of the production-batch loss deviation come from (profiles/r05_parity_seeds.md: slo at C2 sits at
8.1-8.9e-4 for every seed)?  One C2 / C4 step at B = 64: signed per-task loss errors, the loss recomputed in fp32 on the CPU from the
engine's own predictions (isolates the loss kernels from everything upstream), and the scale / offset of the engine's predictions and
encoder tokens against the oracle's (least-squares slope: a truncating conversion shows as a slope below 1).
usage: parity_bias.py c2|c4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullsize_gpu as F
import test_model_gpu as T
from oracle import mirage_oracle as O

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind, doms, B, Kk = ("base", ("bscan", "slo"), 64, 128) if which == "c2" else ("large", ("bscan", "slo", "bscanlayermap"), 64, 98)
DEV = "cuda:0"
cfg = O.make_config(kind, doms)
x = O.synthetic_batch(cfg, B, seed=1234)
(p, u, g), tm, ids_keep, ids_restore = O.synthetic_masks(cfg, B, Kk, seed=99)
noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
model, opt, ts = F._run_engine(cfg, B, Kk, x, noise, use_graph=True)
task = {d: float(v) for d, v in ts.task_losses.items()}
preds = {d: v.float().cpu() for d, v in ts.pred_images().items()}
nS = 8 if which == "c4" else B
idx = torch.arange(nS) * (B // nS)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
sd = O.procedural_state_dict(cfg)
with torch.no_grad():
    xs = {d: t[idx] for d, t in x.items()}
    ref_total, ref_ls, ref_enc, ref_preds = O.forward_loss(cfg, sd, xs, ids_keep[idx], ids_restore[idx])
    tmask = O.task_masks_from_ids(cfg, ids_restore[idx], Kk)
    mine_ls = O.losses(cfg, {d: preds[d][idx] for d in preds}, xs, tmask)
model2, _, _ = T._build(cfg)
model2.train(); model2.output_adapters = None
with torch.no_grad():
    enc, _ = model2({d: x[d].to(DEV) for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
enc = enc.float()[idx.to(DEV)].cpu()

def fit(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    slope = float((a * b).sum() / (b * b).sum())
    return slope, float((a - b).mean()), float((a - b).norm() / b.norm())

print(f"== {which}: B = {B}, oracle on {nS} samples")
for d in doms:
    print(f"{d}: oracle loss {float(ref_ls[d]):.6f}; oracle-loss-code on ENGINE predictions {float(mine_ls[d]):.6f} (rel {(float(mine_ls[d]) - float(ref_ls[d])) / float(ref_ls[d]):+.2e})"
          + (f"; engine's own loss over all {B} samples {task[d]:.6f}" if nS != B else f"; engine's own loss {task[d]:.6f} (rel to oracle {(task[d] - float(ref_ls[d])) / float(ref_ls[d]):+.2e}, rel to CPU-on-engine-preds {(task[d] - float(mine_ls[d])) / float(mine_ls[d]):+.2e})"))
    pr = ref_preds[d].float()
    sl, off, rr = fit(preds[d][idx], pr)
    print(f"   predictions vs oracle: slope {sl:.6f}  mean offset {off:+.3e}  rel-RMS {rr:.2e}   (oracle pred rms {float(pr.double().pow(2).mean().sqrt()):.4f}, mean {float(pr.mean()):+.4f})")
sl, off, rr = fit(enc, ref_enc)
print(f"encoder tokens vs oracle: slope {sl:.6f}  mean offset {off:+.3e}  rel-RMS {rr:.2e}")
