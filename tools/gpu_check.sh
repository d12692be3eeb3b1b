#!/bin/bash
# First-contact run on the GPU box: kernel parity tests, smoke, a short bench.
# Usage (from the repo root, through gpurun):  bash tools/gpu_check.sh [quick|full]
set -u
mkdir -p gpurun_out
MODE=${1:-quick}
export PYTHONUNBUFFERED=1
echo "== rocminfo =="; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
free -g | head -2; nproc
echo "== kernel tests =="
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/kernels.log
echo "== smoke =="
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee gpurun_out/smoke.log
echo "== bench er8192 =="
timeout 900 python bench.py --workload er8192 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_er8192.log
echo "== bench pl32768 =="
timeout 1200 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_pl32768.log
if [ "$MODE" = "full" ]; then
  echo "== parity tests =="
  timeout 3000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/parity.log
fi
