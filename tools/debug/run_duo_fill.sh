#!/bin/bash
# experiment (library built with -DMRG_DUO_EXPERIMENT): MRG_DUO_FILL dummy v_fma_f32 behind every MFMA of the duo kernel's clusters
for f in 0 1 2 3 4 6 8; do
  echo "== MRG_DUO_FILL=$f"
  MRG_DUO_FILL=$f timeout 200 python tools/bench_sk.py fill 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$4,$5, $(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
done
