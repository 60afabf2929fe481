R=$PWD; O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
ls -la $O/trace/*/ $O/pmc_fetch/*/ $O/pmc_write/*/
