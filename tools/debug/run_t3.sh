python -m pytest tests/test_kernels_gpu.py -x -q -k "loss or masked or cross_entropy" 2>&1 | tail -3
python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
python tools/debug/bench_loss.py 2>&1 | grep -v amdgpu
REPS=3 bash tools/ab.sh "c4" "BASE" "MRG_NOP=1"
