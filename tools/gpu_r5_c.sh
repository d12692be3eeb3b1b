#!/bin/bash
# round 5, call C: product-path tests, then the hand-back of a config-4 fit in its forms (second call of each)
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5c2.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_product_path.py -x -q -m gpu 2>&1 | tail -5 >> $O || { tail -40 $O; exit 1; }
for rep in 1 2 3; do
SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan 2>&1 | grep -E "handback_f64|result|total" | tail -3 >> $O
done
echo "symmetric form" >> $O
SIMRANK_SYM_HANDBACK=1 SIMRANK_TIME_HANDBACK=1 timeout -k 10 300 python tools/fit_breakdown.py f32 pl32768d32 plain full plan 2>&1 | grep -E "handback_f64|result|total" | tail -3 >> $O
cat $O
