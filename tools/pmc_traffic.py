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


def per_kernel_grid(dbfile, counter):
    """the same, one row per (kernel, workgroup count): the GEMM families serve several shapes each"""
    db = sqlite3.connect(dbfile)
    rows = db.execute("""select s.display_name, d.grid_size_x / d.workgroup_size_x, count(distinct d.id), sum(e.value)
         from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
         join rocpd_kernel_dispatch d on e.event_id = d.event_id
         join rocpd_info_kernel_symbol s on d.kernel_id = s.id
         where p.name = ? group by s.display_name, d.grid_size_x / d.workgroup_size_x""", (counter,)).fetchall()
    return {(re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0], int(g)): (c, v) for n, g, c, v in rows}


# C2 (Base 512/32 bscan+slo, B = 64: M = 8256 encoder rows) encoder GEMM launches by (kernel family prefix, workgroups): operand bytes
# A + B (+ side operands read by the epilogue) that the launch needs ONCE, and the floor with 8 XCDs whose L2s do not share: every XCD
# reads its row band of A once and ALL of B (B is needed by every row band), i.e. A + 8 B -- the row-band partition the kernels use
# (gemm_common.h wg_remap); a 4 x 2 partition would read 2 A + 4 B, no less for these shapes.
def _mb(*elems):
    return sum(elems) / 1e6
M2, D2 = 8256, 768
C2_GEMMS = {
    ("gemm_glds_kernel<0, 0", 1170): ("qkv forward 8256x2304x768", _mb(M2 * D2 * 2, 2304 * D2 * 2), _mb(M2 * D2 * 2, 8 * 2304 * D2 * 2)),
    ("gemm_glds_kernel<0, 0", 1560): ("fc1 forward 8256x3072x768", _mb(M2 * D2 * 2, 3072 * D2 * 2), _mb(M2 * D2 * 2, 8 * 3072 * D2 * 2)),
    ("gemm_ws256_kernel<0, 0", 198): ("proj / fc2 forward 8256x768x{768,3072} (+ fp32 residual)",
                                      _mb((M2 * D2 * 2 + D2 * D2 * 2 + M2 * D2 * 4) / 2, (M2 * 3072 * 2 + D2 * 3072 * 2 + M2 * D2 * 4) / 2),
                                      _mb((M2 * D2 * 2 + 8 * D2 * D2 * 2 + M2 * D2 * 4) / 2, (M2 * 3072 * 2 + 8 * D2 * 3072 * 2 + M2 * D2 * 4) / 2)),
    ("gemm_glds_kernel<0, 1", 1560): ("fc2 dgrad 8256x3072x768 (+ fp16 GELU' side operand)", _mb(M2 * D2 * 2, 3072 * D2 * 2, M2 * 3072 * 2),
                                      _mb(M2 * D2 * 2, 8 * 3072 * D2 * 2, M2 * 3072 * 2)),
    ("gemm_ws256_kernel<0, 1", 198): ("fc1 / qkv dgrad 8256x768x{3072,2304}", _mb((M2 * 3072 * 2 + 3072 * D2 * 2) / 2, (M2 * 2304 * 2 + 2304 * D2 * 2) / 2),
                                      _mb((M2 * 3072 * 2 + 8 * 3072 * D2 * 2) / 2, (M2 * 2304 * 2 + 8 * 2304 * D2 * 2) / 2)),
}


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
        fg = per_kernel_grid(sys.argv[1], "FETCH_SIZE")
        lines += ["", "## fetch / operand ratio of the encoder GEMM launches (C2)", "",
                  "Fetched = 2 x FETCH_SIZE (the gfx950 correction) per launch; operands = the bytes the launch needs once; 8-L2 floor = what eight",
                  "XCDs whose L2s do not share must fetch with the row-band partition (every XCD: its band of A + all of B).  FETCH_SIZE counts",
                  "requests leaving an XCD's L2, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM section): the distance operands -> floor is",
                  "structural (no multicast between XCDs), the distance floor -> fetched is what rasterisation could still remove.", "",
                  "| launch | kernel (workgroups) | launches | fetched MB | operands MB | fetched / operands | 8-L2 floor MB | fetched / floor |", "|---|---|---|---|---|---|---|---|"]
        for (pref, grid), (what, op_mb, floor_mb) in C2_GEMMS.items():
            hit = [(k, v) for k, v in fg.items() if k[0].startswith(pref) and k[1] == grid]
            if not hit:
                continue
            n = sum(v[0] for _, v in hit); kib = sum(v[1] for _, v in hit) / n
            mb = 2.0 * kib * 1024 / 1e6
            lines.append(f"| {what} | {hit[0][0][0]} ({grid}) | {n} | {mb:.1f} | {op_mb:.1f} | {mb / op_mb:.2f} | {floor_mb:.1f} | {mb / floor_mb:.2f} |")
        open(sys.argv[4], "w").write("\n".join(lines) + "\n")
    print(json.dumps({k: v for k, v in out.items() if "gemm" in k}, indent=1))


if __name__ == "__main__":
    main()
