#!/usr/bin/env python3
"""Average every collected PMC counter per kernel from a rocprofv3 --pmc rocpd database.  Usage: pmc_dump.py results.db [filter]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("""select s.display_name, p.name, count(distinct d.id), sum(e.value)
     from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
     join rocpd_kernel_dispatch d on e.event_id = d.event_id
     join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.display_name, p.name""").fetchall()
for n, c, k, v in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]
    if flt in n:
        print(f"{n[:60]:60s} {c:32s} launches={k:3d} avg={v / k:16.1f}")
