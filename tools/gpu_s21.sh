#!/bin/bash
# after the size rule of the dense sets: the unit sizes again
out=gpurun_out/r04_units_sweep2.log
: > $out
for fu in 24 32 48 64 96; do for fr in 6000 8192 12000; do
  timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_unit=$fu,fuse_rows=$fr >> $out 2>&1 || exit 1
done; done
for fs in 20 28; do for fg in 2 3 4; do
  timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_steps=$fs,fuse_group=$fg >> $out 2>&1 || exit 1
done; done
cat $out
