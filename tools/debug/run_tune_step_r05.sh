#!/bin/bash
# in-step coordinate descent over every kernel variant with the round-5 build (the epilogue wait fix moved the relative speeds)
cd $GRAFT_REPO_ROOT
export MRG_TUNE_TOP=16
for cfg in $1; do
  timeout 2400 python tools/tune_in_step.py $cfg --save > gpurun_out/r05_tune_step_$cfg.log 2>&1
  cp mirage_amd/gemm_tune_gfx950.json gpurun_out/r05_tune_step_after_$cfg.json
  grep -v "trial" gpurun_out/r05_tune_step_$cfg.log | tail -20
done
