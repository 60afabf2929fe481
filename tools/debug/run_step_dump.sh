#!/bin/bash
# kernel trace of a few graph-replayed steps -> flat per-dispatch dump of one step (gpurun_out/step_dump_<cfg>.txt)
CFG=${1:-c2}; R=$PWD; O=$R/gpurun_out/sd; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/db -- python3 $R/bench.py --config $CFG --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $O/log.txt 2>&1
DB=$(ls $O/db/*/*.db | head -1)
python3 $R/tools/rocpd_step_dump.py $DB > $R/gpurun_out/step_dump_$CFG.txt
rm -rf $O/db
