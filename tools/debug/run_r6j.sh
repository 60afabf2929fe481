mkdir -p gpurun_out/r6j
python -m pytest tests/test_model_gpu.py tests/test_rowfuse_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r6j/default.txt
MRG_ROWFUSE=3 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -k "not five_step" 2>&1 | tail -4 > gpurun_out/r6j/rowfuse3.txt
