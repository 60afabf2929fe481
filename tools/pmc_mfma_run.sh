R=$PWD; O=$R/gpurun_out/mfma_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -- python3 $R/bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline > $O/run.log 2>&1
ls -la $O/*/ | tail -2
