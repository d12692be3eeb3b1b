#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s4.log
: > $O
V=$PWD/gpurun_variants
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1 || { tail -30 $O; exit 1; }; }
run "product (early 0, 4 wg)" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2
run "lb3 early 0" env SIMRANK_LIB=$V/libsimrank_hip_f2lb3.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2,fuse_wgs=3
run "lb3 early 1" env SIMRANK_LIB=$V/libsimrank_hip_f2lb3e.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2,fuse_wgs=3
run "lb4 early 1 (spills)" env SIMRANK_LIB=$V/libsimrank_hip_f2e.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2
run "old" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=1
run "stamps" env SIMRANK_LIB=$V/libsimrank_hip_f2st.so timeout -k 10 300 python tools/fused2_stamps.py pl32768d32 --out gpurun_out/r04_f2st.npz
run "stamps cap 1e9" env SIMRANK_LIB=$V/libsimrank_hip_f2st.so timeout -k 10 300 python tools/fused2_stamps.py pl32768d32 --set fuse=2,fuse_cap=1000000000
tail -60 $O
