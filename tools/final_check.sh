#!/bin/bash
# what the driver runs at round end, in one gpurun call: the GPU suite, smoke(), the default bench line (outputs under gpurun_out/final/)
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/final/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
