#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_shardplan.py tests/test_gpu_product_path.py -x -q -m gpu -p no:cacheprovider -k "prior or asym or golden or runs_the_c or refuses" > gpurun_out/asym_a.log 2>&1; tail -15 gpurun_out/asym_a.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "plan_api or biplan or golden_vectors or randomized" > gpurun_out/asym_b.log 2>&1; tail -15 gpurun_out/asym_b.log
