#!/bin/bash
# launch order 0 (matrix-core units in front) against 4 / 8 (spread over the first quarter / eighth), alternating on one box
out=gpurun_out/r04_fuse_order_ab.log
: > $out
for rep in 1 2 3 4; do for fo in 0 4 8; do
  timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 8 --set fuse_order=$fo >> $out 2>&1 || exit 1
done; done
for fo in 0 4; do
  timeout -k 10 200 python tools/leg_only.py --workload pl32768 --steps 8 --set fuse_order=$fo >> $out 2>&1 || exit 1
  timeout -k 10 200 python tools/leg_only.py --workload pl65536 --pp --steps 4 --set fuse_order=$fo >> $out 2>&1 || exit 1
  timeout -k 10 200 python tools/leg_only.py --workload er8192 --steps 20 --set fuse_order=$fo >> $out 2>&1 || exit 1
done
cat $out
