#!/bin/bash
# round 5, call I: fp16-held legs (half.hip) with the prologue's loads untangled: tests, A/B at config 5
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5i.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q -m gpu 2>&1 | tail -4 >> $O || { tail -40 $O; exit 1; }
V=$PWD/gpurun_variants/libsimrank_hip_oldhalf.so
for rep in 1 2; do
  echo -n "old half " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 6 --pp --storage fp16 >> $O 2>&1
  echo -n "new half " >> $O; timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 6 --pp --storage fp16 >> $O 2>&1
done
echo -n "old half " >> $O; SIMRANK_LIB=$V timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --storage fp16 >> $O 2>&1
echo -n "new half " >> $O; timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --storage fp16 >> $O 2>&1
cat $O
