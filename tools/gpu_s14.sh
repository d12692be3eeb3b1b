#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s14.log
: > $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1
echo "pytest rc $?" >> $O; tail -c 600 gpurun_out/r04_gpu_tests.log >> $O
python -c "import __graft_entry__ as g; g.smoke()" >> $O 2>&1
echo "smoke rc $?" >> $O
timeout -k 10 500 python bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err
echo "bench rc $?" >> $O
tail -12 $O
