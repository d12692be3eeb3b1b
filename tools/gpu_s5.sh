#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s5.log
: > $O
V=$PWD/gpurun_variants
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1 || { tail -30 $O; exit 1; }; }
run "old kernel tests (sc1 hand-off of split blocks)" timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu
for fu in 1048576 128 64 32 16; do
run "old fuse_unit=$fu" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=1,fuse_unit=$fu
done
run "old fuse_unit=64 group 2" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=1,fuse_unit=64,fuse_group=2
run "new" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse=2
run "stamps" env SIMRANK_LIB=$V/libsimrank_hip_f2st.so timeout -k 10 300 python tools/fused2_stamps.py pl32768d32 --out gpurun_out/r04_f2st.npz
tail -60 $O
