#!/usr/bin/env python3
"""The masked losses of the step in isolation at the BASELINE shapes (token-layout bf16 predictions): us per launch of the separate
forward / backward launches and of the one-pass form the adapters' lanes use (mrg_masked_loss_fused)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirage_amd import kernels as K, _lib as L
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def t(name, f, reps=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    best = 1e9
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); g.replay(); en.record(); en.synchronize()
        best = min(best, st.elapsed_time(en) / reps * 1e3)
    print(f"{name:50s} {best:7.1f} us", flush=True)
def case(tag, kind, B, C, S, P, frac):
    g = S // P
    pred = torch.randn(B, g * g, C * P * P, device=dev).bfloat16()
    tgt = torch.randint(0, C, (B, S, S), device=dev) if kind == L.MRG_LOSS_CE else torch.rand(B, C, S, S, device=dev)
    wide = (torch.rand(B, 3 * g * g, device=dev) < frac).long()
    mask = wide[:, g * g:2 * g * g]
    d = K.loss_desc(kind, pred, tgt, mask, P)
    loss, coef, part = K.masked_loss_fwd(d, dev)
    t(f"{tag}: forward + finish", lambda: K.masked_loss_fwd(d, dev))
    t(f"{tag}: backward", lambda: K.masked_loss_bwd(d, coef, part, None, pred))
    t(f"{tag}: coefficients from the mask", lambda: K.loss_coef(mask, B, g * g, P * P, dev))
    t(f"{tag}: one pass (value partials + gradient)", lambda: K.masked_loss_fused(d, coef, dev))
case("MSE 512^2 / 32, 1 channel (bscan, slo)", L.MRG_LOSS_MSE, 64, 1, 512, 32, 0.85)
case("CE 128^2 / 8, 13 classes (layer maps)", L.MRG_LOSS_CE, 64, 13, 128, 8, 0.85)
