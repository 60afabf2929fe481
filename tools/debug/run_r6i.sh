mkdir -p gpurun_out/r6i
python -m pytest tests/test_rowfuse_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/r6i/rowfuse_tests.txt
python tools/bench_rowfuse.py > gpurun_out/r6i/bench_rowfuse.txt 2>&1
MRG_LIB_PATH=$PWD/mirage_amd/libmirage_hip_ts.so python tools/debug/rowfuse_ts.py > gpurun_out/r6i/rowfuse_ts.txt 2>&1
REPS=2 bash tools/ab.sh "c2 c4" "MRG_ROWFUSE=0" "MRG_ROWFUSE=2" "MRG_ROWFUSE=3" > gpurun_out/r6i/ab_rowchain.txt 2>&1
