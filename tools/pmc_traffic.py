#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as the TCC
counter slots require).  Units: the counters are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
exactly 2x (MI355X_MICROARCH.md, HBM section), so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Usage: pmc_traffic.py fetch.db write.db out.json [out.md]"""
import json
import re
import sqlite3
import sys


def per_kernel(dbfile, counter):
    db = sqlite3.connect(dbfile)
    rows = db.execute("""select s.display_name, count(distinct d.id), sum(e.value)
         from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
         join rocpd_kernel_dispatch d on e.event_id = d.event_id
         join rocpd_info_kernel_symbol s on d.kernel_id = s.id
         where p.name = ? group by s.display_name""", (counter,)).fetchall()
    return {re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]: (c, v) for n, c, v in rows}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        nf, vf = f.get(k, (0, 0.0)); nw, vw = w.get(k, (0, 0.0))
        fetch_kib = vf / nf if nf else 0.0
        write_kib = vw / nw if nw else 0.0
        out[k] = {"launches": max(nf, nw), "fetch_size_kib_raw": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1),
                  "hbm_bytes_per_launch": int((2.0 * fetch_kib + write_kib) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    if len(sys.argv) > 4:
        lines = ["# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = (2*FETCH + WRITE)*1024)", "",
                 "| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | corrected MB / launch |", "|---|---|---|---|---|"]
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
            lines.append(f"| {k} | {v['launches']} | {v['fetch_size_kib_raw']} | {v['write_size_kib']} | {v['hbm_bytes_per_launch'] / 1e6:.2f} |")
        open(sys.argv[4], "w").write("\n".join(lines) + "\n")
    print(json.dumps({k: v for k, v in out.items() if "gemm" in k}, indent=1))


if __name__ == "__main__":
    main()
