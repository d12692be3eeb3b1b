#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s3.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused2.py -x -q -m gpu 2>&1 | tail -5 >> $O
[ "${PIPESTATUS[0]}" = "0" ] || { tail -30 $O; exit 1; }
for cap in 40000 80000; do
  timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2,fuse_cap=$cap >> $O 2>&1 || { tail -30 $O; exit 1; }
done
echo "lb3 variant" >> $O
for w in 3 4; do
SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_f2lb3.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2,fuse_cap=40000,fuse_wgs=$w >> $O 2>&1
done
timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=1 >> $O 2>&1
timeout -k 10 300 python tools/leg_only.py --workload er8192 --steps 5 --set fuse=2 >> $O 2>&1
timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3 --set fuse=2 >> $O 2>&1
timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3 --set fuse=1 >> $O 2>&1
tail -40 $O
