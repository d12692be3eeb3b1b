#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s13.log
: > $O
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1; }
for kn in "fuse_rows=8192,fuse_unit=32" "fuse_rows=10000,fuse_unit=32" "fuse_rows=12000,fuse_unit=32" "fuse_rows=8192,fuse_unit=24" "fuse_rows=8192,fuse_unit=40" "fuse_rows=8192,fuse_unit=48" "fuse_rows=7000,fuse_unit=32" "fuse_rows=8192,fuse_unit=32,fuse_group=2" "fuse_rows=8192,fuse_unit=32,fuse_group=4" "fuse_rows=8192,fuse_unit=32,fuse_min=4" "fuse_rows=8192,fuse_unit=64"; do
run "$kn" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set $kn
done
run "pl65536 32" timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3 --set fuse_unit=32
run "pl32768 32" timeout -k 10 300 python tools/leg_only.py --workload pl32768 --steps 3 --set fuse_unit=32
run "pl32768 64" timeout -k 10 300 python tools/leg_only.py --workload pl32768 --steps 3 --set fuse_unit=64
grep -v "^==" $O | tail -20
