#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s6.log
: > $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 >> $O
echo "pytest rc ${PIPESTATUS[0]}" >> $O
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r04_a.json 2> gpurun_out/bench_r04_a.err
echo "bench rc $?" >> $O
tail -c 3000 gpurun_out/bench_r04_a.json >> $O
tail -20 $O
