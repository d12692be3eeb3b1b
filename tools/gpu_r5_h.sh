#!/bin/bash
# round 5, call H: the sharded C loops (directed + bipartite) on virtual ranks and over a one-rank RCCL world
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5h.log
: > $O
timeout -k 10 1100 python -m pytest tests/test_gpu_shardplan.py -x -q -m gpu 2>&1 | tail -25 >> $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl" 2>&1 | tail -25 >> $O
cat $O
