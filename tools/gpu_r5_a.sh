#!/bin/bash
# round 5, call A: the new product path (C-level loops behind fit, symmetric hand-back), then the whole GPU suite, then where
# a config-4 fit's wall-clock goes through both solvers
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5a.log
: > $O
free -g | head -2 >> $O; nproc >> $O; cat /sys/kernel/mm/transparent_hugepage/enabled >> $O 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_product_path.py -x -q -m gpu 2>&1 | tail -30 >> $O || { tail -40 $O; exit 1; }
echo "== breakdown plan ==" >> $O
SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan >> $O 2>&1 || { tail -40 $O; exit 1; }
echo "== breakdown python ==" >> $O
timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full python >> $O 2>&1 || { tail -40 $O; exit 1; }
echo "== full suite ==" >> $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r5a_suite.log
tail -5 gpurun_out/r5a_suite.log >> $O
tail -60 $O
