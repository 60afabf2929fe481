python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -3
REPS=3 bash tools/ab.sh "c2 c4" "BASE" "MRG_NOP=1"
