#!/usr/bin/env python3
"""gpurun_out/parity_measured.jsonl (appended by tests/test_model_gpu.py::report during `pytest -m gpu`) -> a markdown table of measured errors
next to the tolerances the tests assert.  Usage: parity_table.py [in.jsonl] [out.md] [title]"""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_measured.jsonl"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/parity_measured.md"
title = sys.argv[3] if len(sys.argv) > 3 else "measured parity errors next to their tolerances (one `pytest -m gpu` run on an MI355X)"
rows, seen = [], set()
for line in open(src):
    d = json.loads(line)
    if "case" not in d:                 # records of recorded skips (e.g. RCCL with two ranks on one GPU) carry no measurements
        continue
    case = d.pop("case")
    tols = {k[:-4]: v for k, v in d.items() if k.endswith("_tol")}
    tol_all = d.get("tol")
    for k, v in d.items():
        if k.endswith("_tol") or k == "tol" or k.endswith("_name") or k in ("B", "oracle_samples"):
            continue
        key = (case, k)
        if key in seen or isinstance(v, (str, bool)) or v is None:
            continue
        if isinstance(v, list):
            if not v or k in ("curve", "ref_curve"):
                continue
            k, v = k + " (max over steps)", max(v)
        seen.add(key)
        base = k.split(".")[0]
        tol = tols.get(k, tols.get(base, tols.get(k.replace("_rel", ""), tol_all)))
        name = d.get(k.replace("_rel", "") + "_name") or d.get(k + "_name")
        label = f"{k} [{name}]" if name else k
        b = f" (B={d['B']})" if "B" in d else ""
        rows.append((case + b, label, v, tol))
with open(dst, "w") as f:
    f.write(f"# {title}\n\nComparator: reference golden vectors (`tests/golden/*.npz`) or the CPU oracle on the same batch / mask noise (`*_trainstep`, `*_8samples`, `dp2_*`).\n"
            "Tolerance column: what the test asserts for that quantity (blank: asserted elsewhere in the test with the value given in its docstring).\n\n"
            "| case | quantity | measured | tolerance |\n|---|---|---|---|\n")
    for case, label, v, tol in rows:
        vs = f"{v:.3g}" if isinstance(v, float) else str(v)
        ts = "" if tol is None else (f"{tol:.3g}" if isinstance(tol, float) else str(tol))
        f.write(f"| {case} | {label} | {vs} | {ts} |\n")
print(len(rows), "rows ->", dst)
