#!/bin/bash
# in-step trial of the duo kernel (variant 13) on every large GEMM shape of a configuration: tools/debug/run_tune_duo.sh "c2 c4 c5"
cd $GRAFT_REPO_ROOT
export MRG_TUNE_TOP=16 MRG_TUNE_ONLY=13 MRG_TUNE_VERBOSE=1
for cfg in $1; do
  timeout 1200 python tools/tune_in_step.py $cfg --save > gpurun_out/r05_tune_duo_$cfg.log 2>&1
  cp mirage_amd/gemm_tune_gfx950.json gpurun_out/r05_tune_after_$cfg.json
  grep -v "trial" gpurun_out/r05_tune_duo_$cfg.log | tail -22
done
