#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s10.log
: > $O
V=$PWD/gpurun_variants
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1 || { tail -30 $O; exit 1; }; }
run "tests" timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu
run "nibble lut" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "byte lut (previous commit)" env SIMRANK_LIB=$V/libsimrank_hip_prevlut.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "nibble lut again" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
for kn in fuse_min=2 fuse_min=4 fuse_group=2 fuse_group=4 fuse_unit=48 fuse_unit=96 fuse_steps=4 fuse_steps=16; do
run "$kn" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set $kn
done
run "pl65536" timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3
run "pl65536 fuse_unit off" timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3 --set fuse_unit=1048576
run "er32768" timeout -k 10 300 python tools/leg_only.py --workload er32768 --steps 3
run "pl32768" timeout -k 10 300 python tools/leg_only.py --workload pl32768 --steps 3
grep -v "^\.\.\." $O | tail -50
