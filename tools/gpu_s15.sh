#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s15.log
: > $O
V=$PWD/gpurun_variants
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/r04_s15_pytest.log 2>&1
echo "pytest rc $?" >> $O; tail -c 300 gpurun_out/r04_s15_pytest.log >> $O
run() { echo "== $1" >> $O; shift; "$@" >> $O 2>&1; }
run "B3" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "B2 (variant)" env SIMRANK_LIB=$V/libsimrank_hip_b2.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "B3" timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "B2 (variant)" env SIMRANK_LIB=$V/libsimrank_hip_b2.so timeout -k 10 300 python tools/leg_only.py --workload pl32768d32 --steps 6
run "B3 pl65536" timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3
run "B2 pl65536" env SIMRANK_LIB=$V/libsimrank_hip_b2.so timeout -k 10 300 python tools/leg_only.py --workload pl65536 --steps 3
run "B3 ml1m" timeout -k 10 300 python tools/bench_cfg3.py
tail -24 $O
