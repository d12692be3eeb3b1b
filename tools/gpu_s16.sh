#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s16.log
: > $O
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1; }
for kn in "fuse_total=0" "fuse_total=150000" "fuse_total=120000" "fuse_total=100000" "fuse_total=80000" "fuse_total=60000" "fuse_total=40000" "fuse_total=100000,fuse_rows=6144" "fuse_total=80000,fuse_unit=32"; do
run "$kn" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set $kn
done
grep -v "^==" $O | tail -12
