#!/usr/bin/env python3
"""One graph-replayed step of a rocprofv3 rocpd database as a flat list: start offset, duration, queue, kernel -- to read
the phase structure (which kernels overlap, where the tails are).  Usage: rocpd_step_dump.py results.db [skip_last]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select s.display_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d
                     join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
ends = [r[2] for r in rows if "step_inc" in r[0]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0, t1 = ends[-2 - skip], ends[-1 - skip]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1 + 1]
qs = {q: i for i, q in enumerate(sorted(set(r[3] for r in sel)))}
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n); n = re.sub(r"\(.*", "", n)
    return n[:70]
print(f"step {(t1 - t0) / 1e3:.1f} us, {len(sel)} dispatches, {len(qs)} queues")
for r in sel:
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} q{qs[r[3]]} wg{r[4] // max(r[5], 1):5d} {short(r[0])}")
