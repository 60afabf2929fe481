mkdir -p gpurun_out/r6f
for m in 0 1; do echo "MRG_LN_BWD_MODE=$m"; MRG_LN_BWD_MODE=$m python tools/bench_ln_bwd.py; done > gpurun_out/r6f/bench_ln_bwd.txt 2>&1
echo "MRG_LN_BWD_MODE=1 MRG_LN_BWD_OCC=3" >> gpurun_out/r6f/bench_ln_bwd.txt; MRG_LN_BWD_OCC=3 python tools/bench_ln_bwd.py >> gpurun_out/r6f/bench_ln_bwd.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or far_negative" 2>&1 | tail -5 > gpurun_out/r6f/ln_tests.txt
REPS=2 bash tools/ab.sh "c2 c4" "MRG_LN_BWD_MODE=0" "MRG_LN_BWD_MODE=1" "MRG_LN_BWD_MODE=1 MRG_WGRAD_LONGK_SPLIT=4" > gpurun_out/r6f/ab_lnbwd.txt 2>&1
