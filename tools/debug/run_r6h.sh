mkdir -p gpurun_out/r6h
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6h/gpu_suite.txt
python bench.py > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/bench.err
