#!/bin/bash
# launch order of a panel's units under the end-of-round plan: matrix-core units in front (0) or spread (k)
out=gpurun_out/r04_fuse_order_sweep.log
: > $out
for fo in 0 1 2 3 4 8; do
  timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_order=$fo >> $out 2>&1 || exit 1
done
for fo in 0 1 2; do
  timeout -k 10 200 python tools/leg_only.py --workload pl65536 --pp --steps 3 --set fuse_order=$fo >> $out 2>&1 || exit 1
done
cat $out
