#!/usr/bin/env python3
"""Hardware-counted MFMA work of one graph-replayed training step from a rocprofv3 --kernel-trace --pmc
SQ_VALU_MFMA_BUSY_CYCLES (GRBM_GUI_ACTIVE) pass.  On gfx950 SQ_VALU_MFMA_BUSY_CYCLES = 32 x (number of 32x32x16 bf16 MFMAs),
summed over all SIMDs, so executed matrix FLOPs = counter x 1024.  Usage: pmc_mfma_step.py results.db ms_per_step algorithmic_TF"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
ms_per_step = float(sys.argv[2]); alg_tf = float(sys.argv[3])
rows = db.execute("""select s.display_name, d.start, d.end, d.id from rocpd_kernel_dispatch d
                     join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
ends = [r[2] for r in rows if "step_inc" in r[0]]
t0, t1 = ends[-3], ends[-2]                       # one graph-replayed step (the last one is bench.py's instrumented eager step)
ids = {r[3]: r[0] for r in rows if r[1] >= t0 and r[2] <= t1 + 1}
pm = db.execute("""select d.id, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                   join rocpd_kernel_dispatch d on e.event_id = d.event_id""").fetchall()
tot = defaultdict(float); per = defaultdict(float)
short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|void ", "", n))[:40]
for did, name, val in pm:
    if did in ids:
        tot[name] += val
        if name == "SQ_VALU_MFMA_BUSY_CYCLES": per[short(ids[did])] += val
busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"]
flops = busy * 1024.0
print(f"# MFMA work of one C2 step from PMC (SQ_VALU_MFMA_BUSY_CYCLES)\n")
print(f"dispatches in the step: {len(ids)}; SQ_VALU_MFMA_BUSY_CYCLES = {busy:.4g} -> {busy / 32:.4g} MFMAs -> {flops / 1e12:.3f} TFLOP executed")
print(f"algorithmic (SURVEY section 8d): {alg_tf:.3f} TFLOP per step -> executed / algorithmic = {flops / 1e12 / alg_tf:.3f} (tile padding, attention key padding)")
print(f"at {ms_per_step:.2f} ms/step (un-profiled bench): executed {flops / ms_per_step / 1e9:.1f} TFLOP/s = {100 * flops / ms_per_step / 1e9 / 2500:.1f} % of the 2.5 PF bf16 peak; "
      f"algorithmic {alg_tf / ms_per_step * 1e3:.1f} TFLOP/s = {100 * alg_tf / ms_per_step * 1e3 / 2500:.1f} %\n")
print("| kernel | MFMA busy cycles / step | share |\n|---|---|---|")
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    if v > 0: print(f"| {k} | {v:.4g} | {100 * v / busy:.1f} % |")
