#!/bin/bash
# kernel tests + leg timings on the standard graphs (quick regression check for kernel work)
set -u
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
[ "${PIPESTATUS[0]}" = "0" ] || exit 1
python tools/sweep.py --workload er:32768:0.00000001,pl32768,er8192,er32768 2>&1 | grep -E "^er|^pl" | cut -c1-22,150-300
python tools/bench_cfg3.py --knobs "dense_min=4,dense_cols=128" 2>&1 | grep "cfg3 mode" | cut -c1-200
