#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s12.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r04_s12_pytest.log 2>&1
echo "pytest rc $?" >> $O; tail -c 1500 gpurun_out/r04_s12_pytest.log >> $O
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1; }
for fr in 1073741824 16384 8192 6144 4096 2048; do
run "fuse_rows=$fr" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set fuse_rows=$fr
done
run "fuse_rows=4096 fuse_unit=32" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set fuse_rows=4096,fuse_unit=32
run "fuse_rows=8192 fuse_unit=32" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6 --set fuse_rows=8192,fuse_unit=32
run "pl65536" timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3
run "er32768" timeout -k 10 300 python tools/leg_only.py --workload er32768 --steps 3
run "er8192" timeout -k 10 300 python tools/leg_only.py --workload er8192 --steps 5
run "ml1m" timeout -k 10 300 python tools/bench_cfg3.py
tail -45 $O
