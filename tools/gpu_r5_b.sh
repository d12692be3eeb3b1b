#!/bin/bash
# round 5, call B: product-path tests, the config-4 fit by phase through both solvers, then the whole bench line
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5b.log
: > $O
cat /sys/fs/cgroup/cpu.max >> $O 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_product_path.py -x -q -m gpu 2>&1 | tail -30 >> $O || { tail -40 $O; exit 1; }
echo "== breakdown plan ==" >> $O
SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan >> $O 2>&1 || { tail -40 $O; exit 1; }
echo "== breakdown plan, symmetric form ==" >> $O
SIMRANK_SYM_HANDBACK=1 SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan 2>&1 | grep -E "handback_f64|result" >> $O
echo "== breakdown plan, 8 / 32 host threads ==" >> $O
SIMRANK_HOST_THREADS=8 SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan 2>&1 | grep handback_f64 >> $O
SIMRANK_HOST_THREADS=32 SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan 2>&1 | grep handback_f64 >> $O
echo "== bench ==" >> $O
timeout -k 10 900 python bench.py > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err || { tail -5 gpurun_out/r5b_bench.err >> $O; }
python tools/bench_summary.py gpurun_out/r5b_bench.json >> $O 2>&1
tail -70 $O
