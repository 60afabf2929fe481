#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 rocpd database for a multi-stream, graph-replayed step:
GPU busy time (union of kernel intervals), idle gaps, concurrency, and per-kernel 'exclusive' time (time during which
the kernel is the only one running) over the last N graph replays.  Usage: rocpd_timeline.py results.db [steps] [skip_last]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cur = db.cursor()
    rows = cur.execute("""select s.display_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    # steps are delimited by the step-counter increment (one per step; AdamW itself is issued in several ranges)
    ends = [r[2] for r in rows if "step_inc" in r[0]]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # trailing steps to ignore (bench.py's instrumented eager step)
    if len(ends) < nsteps + 1 + skip:
        print("not enough steps"); return
    t0, t1 = ends[-nsteps - 1 - skip], ends[-1 - skip]
    sel = [r for r in rows if r[1] >= t0 and r[2] <= t1 + 1]
    span = (t1 - t0) / nsteps
    ev = []
    for i, r in enumerate(sel):
        ev.append((r[1], 1, i)); ev.append((r[2], -1, i))
    ev.sort()
    active = set(); last = t0; busy = 0; conc = defaultdict(int); excl = defaultdict(int); share = defaultdict(float)
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            n = len(active)
            conc[min(n, 4)] += dt
            if n >= 1:
                busy += dt
                for j in active:
                    share[sel[j][0]] += dt / n
                if n == 1:
                    excl[sel[next(iter(active))][0]] += dt
        last = t
        if kind == 1: active.add(i)
        else: active.discard(i)
    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n); n = re.sub(r"\(.*", "", n)
        return n[:60]
    print(f"steps {nsteps}: {span / 1e6:.3f} ms/step; busy {busy / nsteps / 1e6:.3f} ms; idle {(t1 - t0 - busy) / nsteps / 1e6:.3f} ms; "
          f"dispatches/step {len(sel) / nsteps:.0f}; streams {len(set(r[4] for r in sel))} queues {len(set(r[3] for r in sel))}")
    print("concurrency (ms/step): " + ", ".join(f"{k}{'+' if k == 4 else ''}: {v / nsteps / 1e6:.3f}" for k, v in sorted(conc.items())))
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for r in sel:
        a = agg[short(r[0])]; a[2] += 1
    for k, v in share.items(): agg[short(k)][0] += v
    for k, v in excl.items(): agg[short(k)][1] += v
    print("| kernel | calls/step | wall share ms/step | exclusive ms/step |\n|---|---|---|---|")
    for k, (sh, ex, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"| {k} | {n / nsteps:.1f} | {sh / nsteps / 1e6:.3f} | {ex / nsteps / 1e6:.3f} |")


if __name__ == "__main__":
    main()
