timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1; echo "tests rc $?"
tail -12 gpurun_out/r03_gpu_tests.log
