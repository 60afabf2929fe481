mkdir -p gpurun_out/r6c
python -m pytest tests/test_rowfuse_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r6c/rowfuse_tests.txt
python tools/bench_rowfuse.py > gpurun_out/r6c/bench_rowfuse.txt 2>&1
REPS=2 bash tools/ab.sh "c2 c4" "MRG_ROWFUSE=1" "MRG_ROWFUSE=0" > gpurun_out/r6c/ab_rowfuse_fwd.txt 2>&1
python -m pytest tests/test_train_gpu.py -x -q -k "unused_adapter" 2>&1 | tail -40 > gpurun_out/r6c/misc_tests.txt
