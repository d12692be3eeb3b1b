timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_half.py -x -q -k "plan" 2>&1 | tail -2
