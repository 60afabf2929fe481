R=$PWD; O=$R/gpurun_out/r3m; rm -rf $O; mkdir -p $O
python tools/step_phases.py c2 pipelined 2>/dev/null | grep -v amdgpu > $O/phases_new.txt
(cd _ab_head && python tools/step_phases.py c2 pipelined 2>/dev/null | grep -v amdgpu > $O/phases_head.txt)
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/db -- python3 $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/tr.log 2>&1
DB=$(ls $O/db/*/*.db | head -1); python3 $R/tools/rocpd_stats.py $DB $O/kernel_trace.md > /dev/null; rm -rf $O/db
head -8 $O/phases_new.txt; head -8 $O/phases_head.txt; grep -n "maskgen\|embed_plan\|FillFunctor\|gamma\|distribution" $O/kernel_trace.md
