mkdir -p gpurun_out/r6b
python -m pytest tests/test_rowfuse_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r6b/rowfuse_tests.txt
python tools/bench_rowfuse.py > gpurun_out/r6b/bench_rowfuse.txt 2>&1
REPS=2 bash tools/ab.sh "c2 c4" "MRG_ROWFUSE=1" "MRG_ROWFUSE=0" > gpurun_out/r6b/ab_rowfuse_fwd.txt 2>&1
python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -x -q -k "unused_adapter or far_negative or loss" 2>&1 | tail -5 > gpurun_out/r6b/misc_tests.txt
