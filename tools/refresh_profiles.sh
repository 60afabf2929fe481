#!/bin/bash
# Regenerates gpurun_out/refresh/* on an MI355X box: bench lines for every config, kernel trace and HBM-traffic PMC passes of
# the headline (C2) command.  Copy the summaries into profiles/ afterwards (tools/rocpd_stats.py, tools/pmc_traffic.py).
R=$PWD; O=$R/gpurun_out/refresh; mkdir -p $O
timeout 400 python bench.py --config c2 > $O/c2_bench.json 2> $O/c2_bench.err
for c in c3 c4 c5; do timeout 300 python bench.py --config $c > $O/${c}_bench.json 2> /dev/null; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
echo refreshed
