#!/bin/bash
# round-4 session 2: the persistent leg 1 (fuse = 2): correctness, then timings
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s2.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused2.py -x -q -m gpu 2>&1 | tail -15 >> $O
[ "${PIPESTATUS[0]}" = "0" ] || { tail -30 $O; exit 1; }
for cap in 20000 40000 80000 1000000000; do
  timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2,fuse_cap=$cap >> $O 2>&1 || { tail -30 $O; exit 1; }
done
timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=1 >> $O 2>&1
timeout -k 10 300 python tools/leg_only.py --workload er8192 --steps 5 --set fuse=2 >> $O 2>&1
timeout -k 10 300 python tools/leg_only.py --workload er8192 --steps 5 --set fuse=1 >> $O 2>&1
tail -40 $O
