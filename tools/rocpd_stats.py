#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
calls, total / average / min / max duration, share of GPU kernel time.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("""select s.display_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                          group by s.display_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    lines = [f"# rocprofv3 --kernel-trace summary ({sys.argv[1].split('/')[-1]})", "",
             f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; trace span {(span[1] - span[0]) / 1e6:.3f} ms", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, mn, mx in rows:
        lines.append(f"| {short(name)} | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.2f} |")
    # kernel families (template instantiations merged): what bench.py's `roofline.avg_launch_us` is to be compared with
    fam = {}
    for name, n, tot, mn, mx in rows:
        base = re.sub(r"<.*", "", short(name)).split("(")[0]
        f = fam.setdefault(base, [0, 0])
        f[0] += n; f[1] += tot
    lines += ["", "## by kernel family (all template instantiations)", "", "| family | calls | total ms | avg us |", "|---|---|---|---|"]
    for base, (n, tot) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
        lines.append(f"| {base} | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
