timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q -k "odd_sizes or refused" > gpurun_out/half_tests9.log 2>&1; echo "tests rc $?"
tail -12 gpurun_out/half_tests9.log
