#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s8.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half.py -x -q -m gpu 2>&1 | tail -15 >> $O
echo "pytest fused rc ${PIPESTATUS[0]}" >> $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shard or bitwise or logical" 2>&1 | tail -15 >> $O
echo "pytest shards rc ${PIPESTATUS[0]}" >> $O
timeout -k 10 600 python tools/emulate_shards.py pl32768d32 2>&1 | tail -30 >> $O
tail -60 $O
