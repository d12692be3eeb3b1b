#!/bin/bash
# Soaks at HEAD (minutes, outside the test suite): random whole fits, plans and sharded plans against the float64 oracle.
#   usage (through gpurun): bash tools/gpu_soak.sh TAG [fits] [plans] [shardplans]
set -u
TAG=${1:-soak}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python tools/soak_plan.py 0 ${3:-300} > gpurun_out/${TAG}_soak_plan.log 2>&1 || { tail -30 gpurun_out/${TAG}_soak_plan.log; exit 1; }
tail -2 gpurun_out/${TAG}_soak_plan.log
timeout -k 10 500 python tools/soak_shardplan.py 0 ${4:-300} > gpurun_out/${TAG}_soak_shardplan.log 2>&1 || { tail -30 gpurun_out/${TAG}_soak_shardplan.log; exit 1; }
tail -2 gpurun_out/${TAG}_soak_shardplan.log
timeout -k 10 800 python tools/soak_fits.py 100 ${2:-500} > gpurun_out/${TAG}_soak_fits.log 2>&1 || { tail -30 gpurun_out/${TAG}_soak_fits.log; exit 1; }
tail -2 gpurun_out/${TAG}_soak_fits.log
