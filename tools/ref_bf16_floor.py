#!/usr/bin/env python3
"""Noise floor of the REFERENCE itself under bf16 autocast (CPU) on the golden fixtures: how far the reference's own
mixed-precision run is from its fp32 run with identical weights, inputs and masks.  Adds `floor.*` entries to the
fixtures (numbers only), the SIGNED losses of that autocast run (`autocast.loss.*`) and -- round 6 -- `w16.loss.*`: the reference in
fp32 arithmetic with the weight of every nn.Linear / nn.Conv2d MODULE rounded to bf16 (what autocast casts per call), everything else
untouched.  The oracle's `autocast_weights` comparator (a rule on parameter NAMES) must reproduce those numbers
(tests/test_oracle_golden.py).  Build container only."""
import json, os, sys, io, contextlib
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("MIRAGE_REFERENCE", "/root/reference")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import gen_golden as G
from oracle import mirage_oracle as O

def run(name):
    path = os.path.join(G.OUT, name + ".npz")
    z = np.load(path); fx = {k: z[k] for k in z.files}; meta = json.loads(str(fx["meta"]))
    cfg, B, K = meta["cfg"], meta["B"], meta["K"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = G.build_reference(cfg)
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True); model.train()
    x = O.synthetic_batch(cfg, B, seed=meta["data_seed"])
    loss_fns = G.reference_losses(cfg)
    out = {}
    for mode in ("fp32", "bf16", "w16"):
        torch.manual_seed(meta["mask_seed"])
        if mode == "w16":
            with torch.no_grad():
                for m_ in model.modules():
                    if isinstance(m_, (torch.nn.Linear, torch.nn.Conv2d)):
                        m_.weight.copy_(m_.weight.to(torch.bfloat16).to(torch.float32))
        cap = {}
        first = cfg["out_domains"][0]
        orig = model.output_adapters[first].forward
        def spy(encoder_tokens, input_info, ids_keep, ids_restore, orig=orig):
            cap["enc"] = encoder_tokens.detach().float()
            return orig(encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore)
        model.output_adapters[first].forward = spy
        ctxm = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else contextlib.nullcontext()
        with ctxm, contextlib.redirect_stdout(io.StringIO()):
            preds, masks = model({d: x[d] for d in cfg["in_domains"]}, num_encoded_tokens=K, alphas=1.0)
            losses = {d: loss_fns[d](preds[d].float(), x[d], mask=masks.get(d)) for d in preds}
        model.output_adapters[first].forward = orig
        for p_ in model.parameters(): p_.grad = None
        sum(losses.values()).backward()
        out[mode] = dict(losses={d: float(v) for d, v in losses.items()}, enc=cap["enc"],
                         gnorm=float(torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in model.parameters() if p_.grad is not None))),
                         grads={n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None})
    f32, b16 = out["fp32"], out["bf16"]
    for d in f32["losses"]:
        fx[f"autocast.loss.{d}"] = np.float64(b16["losses"][d])
        fx[f"w16.loss.{d}"] = np.float64(out["w16"]["losses"][d])
    fx["w16.grad_total_norm"] = np.float64(out["w16"]["gnorm"])
    for d in f32["losses"]:
        fx[f"floor.loss.{d}"] = np.float64(abs(b16["losses"][d] - f32["losses"][d]) / max(1.0, abs(f32["losses"][d])))
    tot32, tot16 = sum(f32["losses"].values()), sum(b16["losses"].values())
    fx["floor.loss"] = np.float64(abs(tot16 - tot32) / abs(tot32))
    fx["floor.enc_rel_rms"] = np.float64(float((b16["enc"] - f32["enc"]).double().norm() / f32["enc"].double().norm()))
    fx["floor.grad_total_norm"] = np.float64(abs(b16["gnorm"] - f32["gnorm"]) / f32["gnorm"])
    worst = 0.0
    for n, g32 in f32["grads"].items():
        scale = max(float(g32.double().norm()), 1e-3 * f32["gnorm"])
        worst = max(worst, float((b16["grads"][n].double() - g32.double()).norm()) / scale)
    fx["floor.grad_worst_rel"] = np.float64(worst)
    np.savez_compressed(path, **fx)
    print(name, {k: float(v) for k, v in fx.items() if k.startswith("floor.")})

if __name__ == "__main__":
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or ["tiny3", "tiny2", "c2_base", "c3_base", "c4_large"]):
        run(n)
