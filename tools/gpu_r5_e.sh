#!/bin/bash
# round 5, call E: leg 2 (gather3_kernel) with its prologue's loads untangled; A/B against the previous spmm.hip
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5e.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 >> $O || { tail -40 $O; exit 1; }
V=$PWD/gpurun_variants/libsimrank_hip_oldspmm.so
for rep in 1 2 3; do
  echo "-- rep $rep" >> $O
  echo -n "old spmm " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 8 --exact >> $O 2>&1
  echo -n "new spmm " >> $O; timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 8 --exact >> $O 2>&1
done
for w in er8192 pl65536; do
  echo "-- $w" >> $O
  echo -n "old spmm " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload $w --steps 6 --exact >> $O 2>&1
  echo -n "new spmm " >> $O; timeout -k 10 300 python tools/leg_only.py --workload $w --steps 6 --exact >> $O 2>&1
done
echo "-- pl65536 pp" >> $O
echo -n "old spmm " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 6 --pp --exact >> $O 2>&1
echo -n "new spmm " >> $O; timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 6 --pp --exact >> $O 2>&1
echo "-- config 3 + shards" >> $O


cat $O
