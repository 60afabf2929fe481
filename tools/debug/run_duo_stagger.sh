#!/bin/bash
# experiment: start-time stagger / priority of the duo kernel (MRG_GEMM_ABLATE knob) on the fc1 shapes under every epilogue
for a in 0 2 4 6 8 12 256 260 264; do
  echo "== MRG_GEMM_ABLATE=$a"
  MRG_GEMM_ABLATE=$a timeout 200 python tools/bench_sk.py epi 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$4, $(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
done
