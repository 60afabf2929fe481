#!/bin/bash
# Round-6 profile refresh on one MI355X box: the driver's default bench line (C2 headline + C4 / C5 as other_configs), rocprofv3 kernel-trace
# summaries (production schedule = two forward chains, and MRG_TWO_CHAINS=1 = the launch shapes bench.py's instrumented roofline step times),
# device-timestamp phase tables, PMC traffic (with the fetch / operand ratio table) and the MFMA-busy cross-check.  Summaries only: the
# databases are deleted; copy *.md / *.json into profiles/.
R=$PWD; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/r06_bench.json 2> $O/bench.err
MRG_TWO_CHAINS=1 MRG_GEMM_TABLE=1 timeout 300 python bench.py --config c2 --no-cpu-baseline --no-other-configs > $O/r06_c2_bench_one_chain.json 2> $O/c2_gemm_table.err
MRG_GEMM_TABLE=1 timeout 300 python bench.py --config c4 --no-cpu-baseline --no-other-configs > $O/r06_c4_bench.json 2> $O/c4_gemm_table.err
timeout 300 python bench.py --config c3 --no-cpu-baseline --no-other-configs > $O/r06_c3_bench.json 2> /dev/null
for c in c2 c4; do python tools/step_phases.py $c pipelined 2>/dev/null | grep -v amdgpu > $O/r06_${c}_phases.txt; done
cd /tmp; export TMPDIR=/tmp
prof() {  # name config extra-rocprof-args...
  local name=$1 cfg=$2; shift 2
  rm -rf $O/db_$name
  timeout 400 rocprofv3 --kernel-trace "$@" -d $O/db_$name -- python3 $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/$name.log 2>&1
  ls $O/db_$name/*/*.db 2>/dev/null | head -1
}
DB=$(prof c2_trace c2); python3 $R/tools/rocpd_stats.py $DB $O/r06_c2_kernel_trace.md > /dev/null; python3 $R/tools/rocpd_timeline.py $DB 8 > $O/r06_c2_timeline.md; rm -rf $O/db_c2_trace
export MRG_TWO_CHAINS=1
DB=$(prof c2_trace1 c2); python3 $R/tools/rocpd_stats.py $DB $O/r06_c2_kernel_trace_one_chain.md > /dev/null; rm -rf $O/db_c2_trace1
unset MRG_TWO_CHAINS
DB=$(prof c4_trace c4); python3 $R/tools/rocpd_stats.py $DB $O/r06_c4_kernel_trace.md > /dev/null; python3 $R/tools/rocpd_timeline.py $DB 8 > $O/r06_c4_timeline.md; rm -rf $O/db_c4_trace
DB=$(prof c5_trace c5); python3 $R/tools/rocpd_stats.py $DB $O/r06_c5_kernel_trace.md > /dev/null; rm -rf $O/db_c5_trace
export MRG_TWO_CHAINS=1       # traffic per launch of the full-batch launch shapes (what the ratio table is keyed on)
DBF=$(prof c2_fetch c2 --pmc FETCH_SIZE); DBW=$(prof c2_write c2 --pmc WRITE_SIZE)
unset MRG_TWO_CHAINS
python3 $R/tools/pmc_traffic.py $DBF $DBW $O/r06_c2_pmc_traffic.json $O/r06_c2_pmc_traffic.md > /dev/null
python3 - "$DBF" "$DBW" $O/r06_c2_step_traffic.json >> $O/r06_c2_pmc_traffic.md <<'PY'
import re, sqlite3, sys
def step_total(dbfile, counter):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select s.display_name, d.start, d.end, d.id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [r[2] for r in rows if "step_inc" in r[0]]
    t0, t1 = ends[-3], ends[-2]
    ids = {r[3] for r in rows if r[1] >= t0 and r[2] <= t1 + 1}
    pm = db.execute("select d.id, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id where p.name = ?", (counter,)).fetchall()
    return sum(v for i, v in pm if i in ids)
f, w = step_total(sys.argv[1], "FETCH_SIZE"), step_total(sys.argv[2], "WRITE_SIZE")
import json
json.dump({"fetch_raw_gb": f * 1024 / 1e9, "read_gb": 2 * f * 1024 / 1e9, "written_gb": w * 1024 / 1e9, "total_gb": (2 * f + w) * 1024 / 1e9,
           "note": "one graph-replayed C2 step, MRG_TWO_CHAINS=1; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"}, open(sys.argv[3], "w"), indent=1)
print(f"\n## one graph-replayed C2 step (MRG_TWO_CHAINS=1)\n\nFETCH_SIZE {f * 1024 / 1e9:.2f} GB raw, corrected x2 = {2 * f * 1024 / 1e9:.2f} GB read; WRITE_SIZE {w * 1024 / 1e9:.2f} GB written; total {(2 * f + w) * 1024 / 1e9:.2f} GB per step")
PY
rm -rf $O/db_c2_fetch $O/db_c2_write
# the same whole-step totals for C4 (production schedule)
DBF=$(prof c4_fetch c4 --pmc FETCH_SIZE); DBW=$(prof c4_write c4 --pmc WRITE_SIZE)
python3 - "$DBF" "$DBW" $O/r06_c4_step_traffic.json > $O/r06_c4_step_traffic.md <<'PY'
import re, sqlite3, sys, json
def step_total(dbfile, counter):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select s.display_name, d.start, d.end, d.id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [r[2] for r in rows if "step_inc" in r[0]]
    t0, t1 = ends[-3], ends[-2]
    ids = {r[3] for r in rows if r[1] >= t0 and r[2] <= t1 + 1}
    pm = db.execute("select d.id, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id where p.name = ?", (counter,)).fetchall()
    return sum(v for i, v in pm if i in ids)
f, w = step_total(sys.argv[1], "FETCH_SIZE"), step_total(sys.argv[2], "WRITE_SIZE")
json.dump({"fetch_raw_gb": f * 1024 / 1e9, "read_gb": 2 * f * 1024 / 1e9, "written_gb": w * 1024 / 1e9, "total_gb": (2 * f + w) * 1024 / 1e9,
           "note": "one graph-replayed C4 step (production schedule); FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"}, open(sys.argv[3], "w"), indent=1)
print(f"## one graph-replayed C4 step\n\nFETCH_SIZE {f * 1024 / 1e9:.2f} GB raw, corrected x2 = {2 * f * 1024 / 1e9:.2f} GB read; WRITE_SIZE {w * 1024 / 1e9:.2f} GB written; total {(2 * f + w) * 1024 / 1e9:.2f} GB per step")
PY
rm -rf $O/db_c4_fetch $O/db_c4_write
DBM=$(prof c2_mfma c2 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)
MS=$(python3 -c "import json; print(json.load(open('$O/r06_bench.json'))['ms_per_step'])")
python3 $R/tools/pmc_mfma_step.py $DBM $MS 4.94 > $O/r06_c2_mfma_pmc.md; rm -rf $O/db_c2_mfma
rm -f $O/*.log
echo refreshed; ls $O
# the encoder GEMM shapes with their production epilogues (tuned / 8-phase / vendor)
cd $R
python tools/bench_sk.py all 2>/dev/null | grep -v amdgpu > $O/r06_bench_sk.txt
python tools/bench_sk.py epi 2>/dev/null | grep -v amdgpu > $O/r06_bench_epi.txt
ls $O
