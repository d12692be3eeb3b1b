#!/bin/bash
# Block-dense part: kernel tests, then leg timings with the dense part off / on (tools/sweep.py).
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/kernels.log
[ "${PIPESTATUS[0]}" = "0" ] || exit 1
timeout 900 python tools/sweep.py --workload pl32768 --dense-min ${DMIN:-0,2,3,4,6} 2>&1 | tee gpurun_out/sweep_dense.log
