#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s9.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half.py tests/test_gpu_kernels.py tests/test_gpu_fused2.py -x -q -m gpu > gpurun_out/r04_s9_pytest.log 2>&1
echo "pytest rc $?" >> $O
head -c 3000 gpurun_out/r04_s9_pytest.log >> $O; echo ... >> $O; tail -c 1500 gpurun_out/r04_s9_pytest.log >> $O
tail -40 $O
