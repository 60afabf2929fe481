mkdir -p gpurun_out/r6e
REPS=2 bash tools/ab.sh "c2 c4" "MRG_ROWFUSE=0" "MRG_ROWFUSE=1" "MRG_ROWFUSE=2" "MRG_ROWFUSE=3" > gpurun_out/r6e/ab_rowfuse_bits.txt 2>&1
python -m pytest tests/test_train_gpu.py -x -q -k "unused_adapter" 2>&1 | tail -5 > gpurun_out/r6e/misc_tests.txt
python tools/step_phases.py c2 > gpurun_out/r6e/phases_c2_rf0.txt 2>&1
