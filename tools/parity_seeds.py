#!/usr/bin/env python3
"""The production-batch parity cases of tests/test_fullsize_gpu.py (C2 B = 64 graph-replayed TrainStep against the CPU oracle; C4 B = 64 on
8 oracle samples) over several data x mask seeds: how much margin the 1e-3 loss tolerance of BASELINE.json really has (VERDICT r04 "what's
weak" #1: one seed sat at 9.77e-4).  Prints a markdown table; measures only, asserts nothing.

usage: parity_seeds.py [c2] [c4] [--seeds N]      (N x N seed pairs, default 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fullsize_gpu as F

which = [a for a in sys.argv[1:] if a in ("c2", "c4")] or ["c2", "c4"]
n = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else 3
data_seeds = [1234, 20250, 31337, 4242][:n]
mask_seeds = [99, 7, 5150, 808][:n]
rows = []
for cfg in which:
    for ds in data_seeds:
        for ms in mask_seeds:
            if cfg == "c2":
                m = F._fullsize_case("c2_b64_trainstep_seeds", "base", ("bscan", "slo"), 64, 128, check_eager=False, data_seed=ds, mask_seed=ms, asserts=False)
            else:
                m = F._c4_b64_case(data_seed=ds, mask_seed=ms, asserts=False)
            per = {k.split(".", 1)[1]: v for k, v in m.items() if k.startswith("loss_rel.")}
            per16 = {k.split(".", 1)[1]: v for k, v in m.items() if k.startswith("loss_rel_w16.")}
            rows.append((cfg, ds, ms, m["loss_rel"], per, m.get("grad_norm_rel"), m.get("grad_rms_worst"), m["masks_bit_exact"], m["loss_rel_w16"], per16))
            print(f"# {cfg} data {ds} mask {ms}: total {m['loss_rel']:.2e} " + " ".join(f"{k} {v:.2e}" for k, v in per.items()), file=sys.stderr, flush=True)
print("| config | data seed | mask seed | vs fp32 oracle: total | per task | vs oracle with bf16 weight matrices: total | per task | grad-norm rel | worst sampled grad rel-RMS | masks bit-exact |")
print("|---|---|---|---|---|---|---|---|---|---|")
for cfg, ds, ms, lt, per, gn, gr, mb, lt16, per16 in rows:
    print(f"| {cfg} | {ds} | {ms} | {lt:.2e} | " + ", ".join(f"{k} {v:.2e}" for k, v in per.items()) + f" | {lt16:.2e} | " + ", ".join(f"{k} {v:.2e}" for k, v in per16.items())
          + f" | {'' if gn is None else f'{gn:.2e}'} | {'' if gr is None else f'{gr:.2e}'} | {mb} |")
for cfg in which:
    sel = [r for r in rows if r[0] == cfg]
    print(f"\nworst per-task loss_rel at {cfg} over {len(sel)} runs: {max(max(r[4].values()) for r in sel):.2e} against the fp32 oracle, "
          f"{max(max(r[9].values()) for r in sel):.2e} against the oracle with bf16 weight matrices")
