#!/bin/bash
# round 5, call D: leg 1 with the prologue / first matrix-core loads requested early, and the sibling order of the gathered ids;
# A/B against round 4's fused.hip (variant library), interleaved
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5d.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4 >> $O || { tail -40 $O; exit 1; }
V=$PWD/gpurun_variants/libsimrank_hip_r4fused.so
for rep in 1 2 3; do
  echo "-- rep $rep" >> $O
  echo -n "r4 fused      " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 8 >> $O 2>&1
  echo -n "r5 early loads" >> $O; timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 8 >> $O 2>&1

done
for w in pl32768 er8192 pl65536; do
  echo "-- $w" >> $O
  echo -n "r4 fused      " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload $w --steps 6 >> $O 2>&1
  echo -n "r5 early loads" >> $O; timeout -k 10 300 python tools/leg_only.py --workload $w --steps 6 >> $O 2>&1

done
cat $O
