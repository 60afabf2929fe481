R=$PWD; O=$R/gpurun_out/r3t_nolanes; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
MRG_LANES=0 timeout 400 rocprofv3 --kernel-trace -d $O/db -- python3 $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/trace.log 2>&1
DB=$(ls $O/db/*/*.db | head -1)
python3 $R/tools/rocpd_step_dump.py $DB > $O/c2_step_dump.txt 2>&1
rm -rf $O/db
