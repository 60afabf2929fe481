#!/bin/bash
# in-step coordinate descent including the 64-row single-stage kernel (variant 8) and the smaller (output adapter) shapes
cd $GRAFT_REPO_ROOT
export MRG_TUNE_TOP=40 MRG_TUNE_MIN_FLOP=1.5e9
for cfg in $1; do
  timeout 3000 python tools/tune_in_step.py $cfg --save > gpurun_out/r05_tune_stepb_$cfg.log 2>&1
  cp mirage_amd/gemm_tune_gfx950.json gpurun_out/r05_tune_stepb_after_$cfg.json
  grep -v "trial" gpurun_out/r05_tune_stepb_$cfg.log | grep -- "->" | awk '{ if ($0 !~ /\(([0-9]+), ([0-9]+)\) -> \(\1, \2\)/) print }' | tail -20
  grep "final\|baseline" gpurun_out/r05_tune_stepb_$cfg.log
done
