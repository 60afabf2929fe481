cd $GRAFT_REPO_ROOT
export MRG_TUNE_TOP=16
for cfg in c2 c4 c3; do
  timeout 1500 python tools/tune_in_step.py $cfg --save > gpurun_out/r3_tune2_step_$cfg.log 2>&1
  cp mirage_amd/gemm_tune_gfx950.json gpurun_out/r3_tune2_after_$cfg.json
done
