python -m pytest tests/test_kernels_gpu.py -x -q -k "loss or masked" 2>&1 | tail -5
python -m pytest tests/test_train_gpu.py -x -q -k "lanes or smoke or early or fused" 2>&1 | tail -5
python -m pytest tests/test_abi.py -x -q 2>&1 | tail -3
