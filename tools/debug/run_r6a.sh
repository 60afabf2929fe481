mkdir -p gpurun_out/r6a
python -m pytest tests/test_rowfuse_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r6a/rowfuse_tests.txt
python tools/bench_rowfuse.py > gpurun_out/r6a/bench_rowfuse.txt 2>&1
REPS=2 bash tools/ab.sh "c2 c4" "MRG_ROWFUSE=1" "MRG_ROWFUSE=0" "BASE" > gpurun_out/r6a/ab_rowfuse_fwd.txt 2>&1
python -m pytest tests/test_fullsize_gpu.py -x -q -k "c4_large_b64_graph or five_step or c2_base_b64" 2>&1 | tail -30 > gpurun_out/r6a/parity_new.txt
python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r6a/model_kernels.txt
