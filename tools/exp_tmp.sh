timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q > gpurun_out/half_tests5.log 2>&1; echo "tests rc $?"
tail -8 gpurun_out/half_tests5.log
