#!/bin/bash
# dense sets: size rule (fuse_steps) against a density rule (fuse_dens entries per step) on three graphs
out=gpurun_out/r04_fuse_dens_sweep.log
: > $out
for w in pl32768d32 pl32768 "pl65536 --pp"; do
  for cfg in "fuse_steps=8" "fuse_steps=20" "fuse_steps=20,fuse_dens=64" "fuse_steps=20,fuse_dens=80" "fuse_steps=20,fuse_dens=96" "fuse_steps=20,fuse_dens=128" "fuse_steps=32,fuse_dens=96" "fuse_steps=1000,fuse_dens=80" "fuse_steps=1000,fuse_dens=112"; do
    echo -n "$w: " >> $out
    timeout -k 10 200 python tools/leg_only.py --workload $w --steps 4 --set $cfg >> $out 2>&1 || exit 1
  done
done
cat $out
