R=$PWD; O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/r04_bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/db_c5 -- python3 $R/bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/c5.log 2>&1
DB=$(ls $O/db_c5/*/*.db | head -1); python3 $R/tools/rocpd_stats.py $DB $O/r04_c5_kernel_trace.md > /dev/null; rm -rf $O/db_c5
cd $R; python - <<PY
import json
for l in open("gpurun_out/r04c/r04_bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["ms_per_step"], d["value"], d["mfma_frac_of_peak"], {k:(v.get("ms_per_step"), v.get("value"), v.get("whole_step_frac")) for k,v in d.get("other_configs",{}).items()})
PY
