timeout -k 10 900 python tools/soak_kernels.py 1500 2 > gpurun_out/r03_soak.log 2>&1; echo "soak rc $?"
timeout -k 10 900 python tools/soak_kernels.py 1500 3 >> gpurun_out/r03_soak.log 2>&1; echo "soak rc $?"
grep "soak:\|Error\|assert" gpurun_out/r03_soak.log | tail -6
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half.py -x -q 2>&1 | tail -2
