#!/bin/bash
# 4-wave (q4) vs 8-wave (p8) 256x256 kernel: correctness under MRG_P8_Q4=3, then timings of both in separate processes
MRG_P8_Q4=3 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "p8 or grouped" 2>&1 | tail -3
for q in 0 1; do echo "== MRG_P8_Q4=$q"; MRG_P8_Q4=$q python tools/bench_p8.py sq 2>/dev/null | grep -v amdgpu; MRG_P8_Q4=$q python tools/bench_p8.py c5 2>/dev/null | grep -v amdgpu; done
for q in 0 1; do echo "== group-equivalent MRG_P8_Q4=$q"; MRG_P8_Q4=$q python tools/debug/exp_group_wgrad.py 2>/dev/null | grep -v amdgpu | tail -8; done
