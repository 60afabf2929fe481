#!/usr/bin/env python3
"""Idle gaps of a graph-replayed step in a rocprofv3 rocpd database: which (kernel, queue) -> (kernel, queue) transitions
leave the GPU idle, per step.  Usage: rocpd_gaps.py results.db [step_from_end=2]"""
import re, sqlite3, sys
from collections import Counter
db = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = db.execute("""select s.display_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d
                     join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
ends = [r[2] for r in rows if "step_inc" in r[0]]
t0, t1 = ends[-k - 1], ends[-k]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1 + 1]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|void ", "", n))[:34]
ev = sorted([(r[1], 1, i) for i, r in enumerate(sel)] + [(r[2], -1, i) for i, r in enumerate(sel)])
act, last, lasti, gaps = 0, t0, None, []
for t, kk, i in ev:
    if act == 0 and t > last and lasti is not None:
        gaps.append((t - last, lasti, i))
    act += kk; last = t; lasti = i
gaps.sort(reverse=True)
print(f"step {(t1 - t0) / 1e6:.3f} ms, {len(sel)} dispatches, {len(gaps)} gaps, idle {sum(g[0] for g in gaps) / 1e6:.3f} ms; "
      f"gaps > 8 us: {sum(1 for g in gaps if g[0] > 8000)} totalling {sum(g[0] for g in gaps if g[0] > 8000) / 1e6:.3f} ms")
c = Counter(); tot = Counter()
for g, a, b in gaps:
    if g > 8000:
        key = (short(sel[a][0]), sel[a][3], short(sel[b][0]), sel[b][3])
        c[key] += 1; tot[key] += g
for key, v in c.most_common(20):
    print(f"{v:3d} x {tot[key] / v / 1e3:5.1f} us  {key}")
