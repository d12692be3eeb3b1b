timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q -k "plan or refused or needs" > gpurun_out/half_tests7.log 2>&1; echo "tests rc $?"
tail -12 gpurun_out/half_tests7.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "config5 or plan_api" > gpurun_out/half_tests8.log 2>&1; echo "tests rc $?"
tail -6 gpurun_out/half_tests8.log
